// CGContext.cpp -- (target, mode) -> constructor registry behind CGContext.h.
// Behaviour follows reference CGContext.cpp:9-37 (lookup by string compare,
// stderr message + exit(1) when nothing matches, --list in registration order).
#include "CGContext.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace
{
  struct Slot
  {
    const char *target;
    const char *mode;
    CGContext::Factory make;
  };

  // Function-local static: constructed on first use, so Register<> objects in
  // any translation unit can run before or after this file's own statics.
  std::vector<Slot>& registry()
  {
    static std::vector<Slot> slots;
    return slots;
  }
}

void CGContext::add(const char *target, const char *mode, Factory make)
{
  registry().push_back(Slot{target, mode, make});
}

CGContext* CGContext::create(const char *target, const char *mode)
{
  for (const Slot& s : registry())
    if (strcmp(s.target, target) == 0 && strcmp(s.mode, mode) == 0)
      return s.make();

  fprintf(stderr, "\nNo implementation found for %s-%s\n\n", target, mode);
  exit(1);
}

void CGContext::list_contexts()
{
  printf("\nRegistered contexts:\n");
  for (const Slot& s : registry())
    printf("\t%s-%s\n", s.target, s.mode);
  printf("\n");
}

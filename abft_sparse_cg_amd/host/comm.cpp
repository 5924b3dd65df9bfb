// comm.cpp -- see comm.h.  Plain POSIX sockets; nothing here touches the GPU.
#include "comm.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static void die(const char *what)
{
  fflush(stdout);
  fprintf(stderr, "hip backend (comm): %s: %s\n", what, strerror(errno));
  exit(2);
}

static const int kMagic = 0x41424654;  // "ABFT"

static int env_int(const char *name, int fallback)
{
  const char *v = getenv(name);
  return v && *v ? atoi(v) : fallback;
}

// ABFT_HIP_GPUS=N without a launcher: the process starts the other N - 1 ranks itself, before
// anything has touched a GPU (plain fork: every rank then runs the unchanged driver from where
// the backend was created; SURVEY 5 names the variable).  Returns the children's pids.
static std::vector<int> g_children;
static int g_listen_fd = -1;  // self-launch: rank 0's rendezvous socket, bound before the fork and kept

// every socket this process holds towards the other ranks (rank 0: one per peer + the listening one);
// kept here so that the exit handler below can close them: file descriptors otherwise outlive the handlers
static std::vector<int> g_open_fds;

static void forget_fd(int fd)
{
  for (size_t i = 0; i < g_open_fds.size(); i++)
    if (g_open_fds[i] == fd) { g_open_fds.erase(g_open_fds.begin() + i); return; }
}

static int status_of(int status)
{
  return WIFEXITED(status) ? WEXITSTATUS(status) : WIFSIGNALED(status) ? 128 + WTERMSIG(status) : 1;
}

// wait for the ranks this process started, at most `limit_s` seconds; the ones still running then get
// SIGTERM, two seconds later SIGKILL.  -> the first non-zero exit status among them (128 + signal)
static int reap_children(int limit_s)
{
  int worst = 0;
  timeval t0;
  gettimeofday(&t0, NULL);
  int stage = 0;  // 0: waiting, 1: SIGTERM sent, 2: SIGKILL sent
  while (!g_children.empty())
  {
    for (size_t i = 0; i < g_children.size();)
    {
      int status = 0;
      const pid_t got = waitpid((pid_t)g_children[i], &status, WNOHANG);
      if (got == 0) { i++; continue; }
      if (got > 0 && status_of(status) && !worst) worst = status_of(status);
      g_children.erase(g_children.begin() + i);  // reaped, or not ours to wait for any more
    }
    if (g_children.empty()) break;
    timeval now;
    gettimeofday(&now, NULL);
    const double waited = (now.tv_sec - t0.tv_sec) + 1e-6 * (now.tv_usec - t0.tv_usec);
    if (stage < 2 && waited > limit_s + 2.0 * stage)
    {
      for (size_t i = 0; i < g_children.size(); i++) kill((pid_t)g_children[i], stage == 0 ? SIGTERM : SIGKILL);
      if (!worst) worst = 128 + (stage == 0 ? SIGTERM : SIGKILL);
      stage++;
    }
    usleep(2000);
  }
  return worst;
}

// Runs on every exit() of the rank that self-launched the others (the driver's normal return, fail(),
// the fatal-ECC exit(1), check(), a rendezvous that timed out): the children are always reaped, and a
// rank that crashed or exited non-zero is not hidden behind a parent that returns 0.  On a failing exit
// the sockets are shut first -- descriptors close only after the handlers, and a child blocked in a
// recv from this rank would otherwise never see the end of the stream while this rank waits for it --
// and the wait is short; on a clean one the children are finishing the same program and get the
// rendezvous' own deadline.
static void reap_at_exit(int status, void *)
{
  if (status != 0)
  {
    for (size_t i = 0; i < g_open_fds.size(); i++) shutdown(g_open_fds[i], SHUT_RDWR), close(g_open_fds[i]);
    g_open_fds.clear();
  }
  const int worst = reap_children(status != 0 ? 5 : env_int("ABFT_COMM_TIMEOUT", 120));
  if (status == 0 && worst != 0)
  {
    fprintf(stderr, "hip backend (comm): a rank started by ABFT_HIP_GPUS ended with status %d\n", worst);
    fflush(NULL);
    _exit(worst);
  }
}

static void self_launch(int n)
{
  // fork() is only safe while nothing has initialised the GPU runtime.  Under rocprofv3 the preloaded
  // tool library has done so before main(): the forked ranks would inherit an unusable runtime.
  const char *preload = getenv("LD_PRELOAD");
  if (getenv("ROCP_TOOL_LIBRARIES") || (preload && (strstr(preload, "rocprof") || strstr(preload, "roctracer"))))
  {
    fprintf(stderr, "hip backend (comm): ABFT_HIP_GPUS cannot fork its ranks under a profiler's preloaded library "
            "(the GPU runtime is already initialised); start the ranks with a launcher instead: "
            "host/mgpu-run N -- rocprofv3 ... -- cg-csr ...\n");
    exit(2);
  }
  // the rendezvous socket: bound once here, on a port the kernel picks, and kept by rank 0 (nothing
  // is probed and released again for another process to grab)
  int port = 29400;
  g_listen_fd = socket(AF_INET, SOCK_STREAM, 0);
  if (g_listen_fd < 0) die("socket");
  g_open_fds.push_back(g_listen_fd);
  {
    const int one = 1;
    setsockopt(g_listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in sa;
    memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET;
    sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    socklen_t len = sizeof(sa);
    if (bind(g_listen_fd, (sockaddr *)&sa, sizeof(sa)) < 0 || getsockname(g_listen_fd, (sockaddr *)&sa, &len) < 0 ||
        listen(g_listen_fd, n) < 0)
      die("bind/listen (self-launch rendezvous)");
    port = ntohs(sa.sin_port);
  }
  char buf[32];
  setenv("MASTER_ADDR", "127.0.0.1", 1);
  snprintf(buf, sizeof(buf), "%d", port);
  setenv("MASTER_PORT", buf, 1);
  snprintf(buf, sizeof(buf), "%d", n);
  setenv("WORLD_SIZE", buf, 1);
  setenv("RANK", "0", 1);
  setenv("LOCAL_RANK", "0", 1);
  fflush(stdout);
  fflush(stderr);
  for (int r = 1; r < n; r++)
  {
    const pid_t pid = fork();
    if (pid < 0) die("fork");
    if (pid == 0)
    {
      snprintf(buf, sizeof(buf), "%d", r);
      setenv("RANK", buf, 1);
      setenv("LOCAL_RANK", buf, 1);
      g_children.clear();
      forget_fd(g_listen_fd);
      close(g_listen_fd);  // rank 0's
      g_listen_fd = -1;
      return;
    }
    g_children.push_back((int)pid);
  }
  on_exit(reap_at_exit, NULL);
}

Comm* Comm::from_env()
{
  if (env_int("WORLD_SIZE", 0) == 0 && env_int("ABFT_HIP_GPUS", 1) > 1)
    self_launch(env_int("ABFT_HIP_GPUS", 1));
  const int world = env_int("WORLD_SIZE", 1);
  // (ABFT_COMM_FORCE=1: take the partitioned code path -- and RCCL -- with a single rank, for tests)
  if (world <= 1 && env_int("ABFT_COMM_FORCE", 0) == 0)
    return NULL;
  Comm *c = new Comm();
  c->size_ = world < 1 ? 1 : world;
  c->rank_ = env_int("RANK", 0);
  c->local_rank_ = env_int("LOCAL_RANK", c->rank_);
  if (c->rank_ < 0 || c->rank_ >= world)
  {
    fprintf(stderr, "hip backend (comm): RANK %d outside [0,%d)\n", c->rank_, world);
    exit(2);
  }
  const char *addr = getenv("MASTER_ADDR");
  // ABFT_COMM_PORT if given; else MASTER_PORT -- plus one under torch.distributed.run, whose agent
  // keeps its own store listening on MASTER_PORT itself
  int port = env_int("MASTER_PORT", 29400);
  const char *agent = getenv("TORCHELASTIC_USE_AGENT_STORE");
  if (agent && !strcmp(agent, "True"))
    port += 1;
  port = env_int("ABFT_COMM_PORT", port);
  c->connect_star(addr && *addr ? addr : "127.0.0.1", port);
  return c;
}

Comm::~Comm()
{
  if (rccl_)
    abft_rccl_destroy(rccl_);
  for (size_t i = 0; i < peers_.size(); i++)
    if (peers_[i] >= 0)
    {
      forget_fd(peers_[i]);
      close(peers_[i]);
    }
  if (listen_fd_ >= 0)
  {
    forget_fd(listen_fd_);
    close(listen_fd_);
  }
  // (ABFT_HIP_GPUS: the ranks this process started are reaped, and their status kept, by reap_at_exit)
}

int Comm::rccl_comm_count(int *device) const { return abft_rccl_comm_count(rccl_, device); }

void Comm::enable_device_collectives(int device)
{
  const char *mode = getenv("ABFT_COMM");
  if (mode && !strcmp(mode, "tcp"))
    return;
  rccl_ = abft_rccl_init(this, device);  // NULL when built without RCCL
}

// Rank 0 listens and accepts size-1 connections; every other rank connects
// (retrying while rank 0 is still starting) and introduces itself by its rank.
void Comm::connect_star(const char *addr, int port)
{
  const int one = 1;
  if (rank_ == 0)
  {
    if (g_listen_fd >= 0)
    {
      listen_fd_ = g_listen_fd;  // self-launch: already bound and listening
      g_listen_fd = -1;
    }
    else
    {
      listen_fd_ = socket(AF_INET, SOCK_STREAM, 0);
      if (listen_fd_ < 0) die("socket");
      g_open_fds.push_back(listen_fd_);
      setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
      sockaddr_in sa;
      memset(&sa, 0, sizeof(sa));
      sa.sin_family = AF_INET;
      sa.sin_port = htons((uint16_t)port);
      // one node: listen on the loopback address when that is what the ranks were given
      in_addr local;
      sa.sin_addr.s_addr = htonl(INADDR_ANY);
      if (inet_pton(AF_INET, addr, &local) == 1 && (ntohl(local.s_addr) >> 24) == 127)
        sa.sin_addr = local;
      if (bind(listen_fd_, (sockaddr *)&sa, sizeof(sa)) < 0) die("bind (MASTER_PORT in use?)");
      if (listen(listen_fd_, size_) < 0) die("listen");
    }
    peers_.assign(size_, -1);
    // a peer that died before connecting (bad device, failed exec) must not leave rank 0
    // -- and the launcher waiting on it -- blocked for ever: one deadline for the rendezvous
    const int limit_s = env_int("ABFT_COMM_TIMEOUT", 120);
    timeval t0;
    gettimeofday(&t0, NULL);
    for (int k = 1; k < size_; k++)
    {
      timeval now;
      gettimeofday(&now, NULL);
      const long left_ms = (long)limit_s * 1000 - ((now.tv_sec - t0.tv_sec) * 1000 + (now.tv_usec - t0.tv_usec) / 1000);
      pollfd pf = {listen_fd_, POLLIN, 0};
      if (left_ms <= 0 || poll(&pf, 1, (int)left_ms) <= 0)
      {
        fflush(stdout);
        fprintf(stderr, "hip backend (comm): rendezvous timed out after %d s; missing rank(s):", limit_s);
        for (int r = 1; r < size_; r++)
          if (peers_[r] < 0) fprintf(stderr, " %d", r);
        fprintf(stderr, "\n");
        exit(2);
      }
      int fd = accept(listen_fd_, NULL, NULL);
      if (fd < 0) die("accept");
      g_open_fds.push_back(fd);
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      timeval hello_to = {10, 0};  // the hello follows the connect at once
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &hello_to, sizeof(hello_to));
      int hello[2] = {0, -1};  // {magic, rank}
      recv_all(fd, hello, sizeof(hello));
      timeval none = {0, 0};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof(none));
      const int who = hello[1];
      if (hello[0] != kMagic || who <= 0 || who >= size_ || peers_[who] >= 0)
      {
        fprintf(stderr, "hip backend (comm): unexpected peer at rendezvous (rank %d)\n", who);
        exit(2);
      }
      send_all(fd, &kMagic, sizeof(kMagic));
      peers_[who] = fd;
    }
    // everyone is here: stop listening, so that a rank of a job started right behind this one
    // (bench.py runs two in a row on the same port) is refused and retries until ITS rank 0 is up,
    // instead of queueing on a socket nobody will accept from again
    forget_fd(listen_fd_);
    close(listen_fd_);
    listen_fd_ = -1;
    return;
  }
  sockaddr_in sa;
  memset(&sa, 0, sizeof(sa));
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1)
  {
    fprintf(stderr, "hip backend (comm): MASTER_ADDR '%s' is not an IPv4 address\n", addr);
    exit(2);
  }
  int fd = -1;
  for (int attempt = 0; attempt < 600; attempt++)  // up to ~60 s for rank 0 to come up
  {
    fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) die("socket");
    if (connect(fd, (sockaddr *)&sa, sizeof(sa)) == 0)
      break;
    close(fd);
    fd = -1;
    usleep(100 * 1000);
  }
  if (fd < 0) die("connect to rank 0");
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  // introduce ourselves and make sure it is rank 0 of this job that answered (not some other
  // service on MASTER_PORT): the reply must arrive, and be the magic word, within a minute
  int hello[2] = {kMagic, rank_};
  send_all(fd, hello, sizeof(hello));
  timeval tv;
  tv.tv_sec = 60; tv.tv_usec = 0;
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  int ack = 0;
  recv_all(fd, &ack, sizeof(ack));
  if (ack != kMagic)
  {
    fprintf(stderr, "hip backend (comm): MASTER_PORT %d answered, but not as rank 0 of this job\n", port);
    exit(2);
  }
  tv.tv_sec = 0;
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  peers_.assign(1, fd);
}

void Comm::send_all(int fd, const void *buf, size_t n)
{
  const char *p = (const char *)buf;
  while (n)
  {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0)
    {
      if (errno == EINTR) continue;
      die("send (a peer ended early?)");
    }
    p += k;
    n -= (size_t)k;
  }
}

void Comm::recv_all(int fd, void *buf, size_t n)
{
  char *p = (char *)buf;
  while (n)
  {
    ssize_t k = recv(fd, p, n, 0);
    if (k < 0)
    {
      if (errno == EINTR) continue;
      die("recv");
    }
    if (k == 0)
    {
      // a peer that exits (e.g. with status 1 on a fatal ECC event it found first) closes its socket
      fflush(stdout);
      fprintf(stderr, "hip backend (comm): rank %d lost a peer\n", rank_);
      exit(3);
    }
    p += k;
    n -= (size_t)k;
  }
}

void Comm::bcast(void *buf, size_t bytes, int root)
{
  if (rank_ == 0)
  {
    if (root != 0)
      recv_all(peers_[root], buf, bytes);
    for (int k = 1; k < size_; k++)
      if (k != root)
        send_all(peers_[k], buf, bytes);
  }
  else if (rank_ == root)
    send_all(peers_[0], buf, bytes);
  else
    recv_all(peers_[0], buf, bytes);
}

void Comm::allgather(const void *mine, size_t bytes, void *all)
{
  char *out = (char *)all;
  if (rank_ == 0)
  {
    memcpy(out, mine, bytes);
    for (int k = 1; k < size_; k++)
      recv_all(peers_[k], out + (size_t)k * bytes, bytes);
    for (int k = 1; k < size_; k++)
      send_all(peers_[k], out, (size_t)size_ * bytes);
  }
  else
  {
    send_all(peers_[0], mine, bytes);
    recv_all(peers_[0], out, (size_t)size_ * bytes);
  }
}

void Comm::allgatherv(const void *mine, size_t bytes, std::vector<char> &all, std::vector<size_t> &sizes)
{
  std::vector<unsigned long long> n(size_);
  unsigned long long me = bytes;
  allgather(&me, sizeof(me), n.data());
  sizes.assign(n.begin(), n.end());
  size_t total = 0;
  for (int k = 0; k < size_; k++) total += sizes[k];
  all.resize(total ? total : 1);
  if (rank_ == 0)
  {
    size_t off = 0;
    for (int k = 0; k < size_; k++)
    {
      if (k == 0) { if (bytes) memcpy(all.data(), mine, bytes); }
      else if (sizes[k]) recv_all(peers_[k], all.data() + off, sizes[k]);
      off += sizes[k];
    }
    if (total)
      for (int k = 1; k < size_; k++)
        send_all(peers_[k], all.data(), total);
  }
  else
  {
    if (bytes) send_all(peers_[0], mine, bytes);
    if (total) recv_all(peers_[0], all.data(), total);
  }
  all.resize(total);
}

void Comm::allreduce_sum(double *v, int n)
{
  if (rank_ == 0)
  {
    std::vector<double> in(n);
    for (int k = 1; k < size_; k++)  // fixed order: rank 1, 2, ... onto rank 0's values
    {
      recv_all(peers_[k], in.data(), sizeof(double) * n);
      for (int i = 0; i < n; i++) v[i] += in[i];
    }
    for (int k = 1; k < size_; k++)
      send_all(peers_[k], v, sizeof(double) * n);
  }
  else
  {
    send_all(peers_[0], v, sizeof(double) * n);
    recv_all(peers_[0], v, sizeof(double) * n);
  }
}

void Comm::barrier()
{
  double z = 0.0;
  allreduce_sum(&z, 1);
}

// Rank 0 relays: it first takes every other rank's outgoing pieces (a header of
// {dst, bytes} pairs, then the payloads), then hands each rank what is addressed to it,
// sources in ascending order -- the order in which the receiver lists its `in` pieces.
void Comm::exchange(const std::vector<Piece> &out, const std::vector<Piece> &in)
{
  struct Item { int src, dst; std::vector<char> data; };
  if (rank_ != 0)
  {
    unsigned long long n = out.size();
    send_all(peers_[0], &n, sizeof(n));
    for (size_t k = 0; k < out.size(); k++)
    {
      unsigned long long head[2] = {(unsigned long long)out[k].peer, out[k].bytes};
      send_all(peers_[0], head, sizeof(head));
      if (out[k].bytes) send_all(peers_[0], out[k].buf, out[k].bytes);
    }
    for (size_t k = 0; k < in.size(); k++)
      if (in[k].bytes) recv_all(peers_[0], in[k].buf, in[k].bytes);
    return;
  }
  std::vector<Item> items;
  for (size_t k = 0; k < out.size(); k++)
  {
    Item it;
    it.src = 0; it.dst = out[k].peer;
    it.data.assign((const char *)out[k].buf, (const char *)out[k].buf + out[k].bytes);
    items.push_back(it);
  }
  for (int r = 1; r < size_; r++)
  {
    unsigned long long n = 0;
    recv_all(peers_[r], &n, sizeof(n));
    for (unsigned long long k = 0; k < n; k++)
    {
      unsigned long long head[2];
      recv_all(peers_[r], head, sizeof(head));
      Item it;
      it.src = r; it.dst = (int)head[0];
      it.data.resize(head[1]);
      if (head[1]) recv_all(peers_[r], it.data.data(), head[1]);
      items.push_back(it);
    }
  }
  // items are in ascending source order already; deliver per destination in that order
  for (int d = 0; d < size_; d++)
  {
    size_t k_in = 0;
    for (size_t i = 0; i < items.size(); i++)
    {
      if (items[i].dst != d) continue;
      if (d == 0)
      {
        if (k_in >= in.size() || in[k_in].peer != items[i].src || in[k_in].bytes != items[i].data.size())
        {
          fprintf(stderr, "hip backend (comm): exchange pieces do not pair up\n");
          exit(2);
        }
        if (in[k_in].bytes) memcpy(in[k_in].buf, items[i].data.data(), in[k_in].bytes);
        k_in++;
      }
      else if (!items[i].data.empty())
        send_all(peers_[d], items[i].data.data(), items[i].data.size());
    }
  }
}

void Comm::allreduce_sum_device(double *dev, int n, void *stream)
{
  abft_rccl_allreduce_sum(rccl_, dev, n, stream);
}

void Comm::device_exchange_begin(void *stream, bool beside) { abft_rccl_exchange_begin(rccl_, stream, beside); }
void Comm::device_exchange_finish(void *stream) { abft_rccl_exchange_finish(rccl_, stream); }

void Comm::allgather_device(double *full, size_t slot)
{
  abft_rccl_allgather(rccl_, full, slot, rank_);
}

void Comm::sendrecv_device(const std::vector<Piece> &out, const std::vector<Piece> &in)
{
  abft_rccl_sendrecv(rccl_, out, in);
}

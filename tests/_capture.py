"""Run a callable in a forked child with C-level stdout captured.

The reference backend reports ECC events with printf() and ends the process
with exit(1) on fatal ones; forking lets a test observe both without dying."""
import os
import pickle
import tempfile


def run_captured(fn, *args, **kw):
    """-> (exit_code, stdout_text, result or None).  `result` is fn's return
    value (pickled through a pipe) when the child got that far."""
    out = tempfile.TemporaryFile()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 0
        try:
            os.close(r)
            os.dup2(out.fileno(), 1)
            try:  # the reference may legitimately segfault (UB paths): keep it quiet
                import faulthandler
                faulthandler.disable()
            except Exception:
                pass
            res = fn(*args, **kw)
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            with os.fdopen(w, "wb") as f:
                pickle.dump(res, f)
        except SystemExit as e:
            code = int(e.code or 0)
        except BaseException:
            import traceback
            traceback.print_exc()
            code = 99
        os._exit(code)
    os.close(w)
    with os.fdopen(r, "rb") as f:
        data = f.read()
    _, status = os.waitpid(pid, 0)
    code = os.WEXITSTATUS(status) if os.WIFEXITED(status) else -os.WTERMSIG(status)
    out.seek(0)
    text = out.read().decode(errors="replace")
    out.close()
    res = pickle.loads(data) if data else None
    return code, text, res

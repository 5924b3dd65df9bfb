"""Run a reference-side helper in a clean forked child with C-level stdout captured.

The reference backend reports ECC events with printf() and ends the process
with exit(1) on fatal ones, so a test can only observe it from outside.  The
helpers (top-level functions of tests/_oracle.py, named ref_*) run in children
forked from a small dedicated server process -- a fresh interpreter that has
imported nothing but numpy and _oracle -- rather than from the pytest process:
forking a process that has live OpenMP / BLAS thread pools deadlocks or crawls.
"""
import atexit
import os
import pickle
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
_server = None


def _send(f, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(data)))
    f.write(data)
    f.flush()


def _recv(f):
    hdr = f.read(8)
    if len(hdr) < 8:
        raise EOFError
    (n,) = struct.unpack("<Q", hdr)
    return pickle.loads(f.read(n))


def _start():
    global _server
    if _server is None or _server.poll() is not None:
        env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        _server = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--serve"], stdin=subprocess.PIPE,
                                   stdout=subprocess.PIPE, cwd=HERE, env=env)
        atexit.register(_stop)
    return _server


def _stop():
    global _server
    if _server is not None and _server.poll() is None:
        try:
            _server.stdin.close()
            _server.wait(5)
        except Exception:
            _server.kill()
    _server = None


def run_captured(fn, *args):
    """fn: a top-level function of tests/_oracle.py.  -> (exit_code, stdout_text,
    result or None); result is fn's return value when the child got that far."""
    srv = _start()
    _send(srv.stdin, (fn.__name__, args))
    return _recv(srv.stdout)


def _serve():
    import _oracle
    inp, out = sys.stdin.buffer, os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)  # stray prints of the server itself go to stderr, not into the protocol
    while True:
        try:
            name, args = _recv(inp)
        except EOFError:
            return
        cap = tempfile.TemporaryFile()
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 0
            try:
                os.close(r)
                os.dup2(cap.fileno(), 1)
                res = getattr(_oracle, name)(*args)
                import ctypes
                ctypes.CDLL(None).fflush(None)
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
        cap.seek(0)
        text = cap.read().decode(errors="replace")
        cap.close()
        _send(out, (code, text, pickle.loads(data) if data else None))


if __name__ == "__main__" and "--serve" in sys.argv:
    sys.path.insert(0, HERE)
    _serve()

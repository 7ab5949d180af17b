"""Where the tests put the programs they compile and the files those write: /tmp/ -- or, under pytest-xdist, a directory per worker, so
that two workers do not compile to, execute and overwrite the same file (seen as "Text file busy" / "Permission denied" when the GPU
suite was run with -n 5)."""
import os

_w = os.environ.get("PYTEST_XDIST_WORKER", "")
TMPW = "/tmp/" if not _w else "/tmp/" + "weldacs_xdist_%d_%s/" % (os.getuid(), _w)
os.makedirs(TMPW, exist_ok=True)

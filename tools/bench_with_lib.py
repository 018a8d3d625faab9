"""Runs bench.py against another build of the library (A/B of a development variant): python tools/bench_with_lib.py LIB [bench args...]"""
import runpy, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from matchtigs_amd import _lib
_lib.LIB_PATH = type(_lib.LIB_PATH)(str(Path(sys.argv[1]).resolve()))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(str(Path(__file__).resolve().parent.parent / "bench.py"), run_name="__main__")

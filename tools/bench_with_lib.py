"""Run bench.py against an alternative build of the library: TACO_LIB=path python tools/bench_with_lib.py [bench args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')

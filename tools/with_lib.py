import os, sys, runpy
sys.path.insert(0, os.getcwd())
from uvc_amd import _lib
if os.environ.get("UVC_LIB"):
    _lib.LIB_PATH = os.environ["UVC_LIB"]
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

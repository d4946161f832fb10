"""Development helper: bench.py over ANOTHER build of the library.  usage: python tools/ab_bench.py <lib.so> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import devlib
sys.modules["video-annotator_amd"] = devlib.load(sys.argv[1])
import bench
sys.argv = ["bench.py"] + sys.argv[2:]
sys.exit(bench.main())

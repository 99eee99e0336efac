import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "90")), exit=True)
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()

"""Runs bench.main() with a watchdog that dumps all Python stacks if it takes too long."""
import faulthandler, sys, os
faulthandler.dump_traceback_later(35, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.main()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, "/root/repo")
from eco_amd import hip
hip.LIB_PATH = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
sys.path.insert(0, "/root/repo/tools")
import conv_bench
conv_bench.main()

import os, sys, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_subbatch as b
b.one("goldie", "q4_0", 64, 8, 32)
b.one("goldie", "q4_0", 64, 1000, 16)
b.one("nano", "q8_0", 64, 8, 32)

"""Developer tool: a few 64-stream batched decode steps (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_modes as b
tier = sys.argv[1] if len(sys.argv) > 1 else "goldie"
b.batch(tier, sys.argv[2] if len(sys.argv) > 2 else "q4_0", int(sys.argv[3]) if len(sys.argv) > 3 else 64, steps=8)

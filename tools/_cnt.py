import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanollama_amd import _lib
print("rank", os.environ.get("RANK"), "count", _lib.lib().nl_device_count(), {k: v for k, v in os.environ.items() if "VISIBLE" in k or k.startswith("HSA") or k.startswith("ROCR")})

ulimit -c 0
cd $GRAFT_REPO_ROOT
python -c "
from nanollama_amd import _lib; import time, os
print('solo count', _lib.lib().nl_device_count())"
(python -c "
from nanollama_amd import _lib; import time
print('A count', _lib.lib().nl_device_count()); time.sleep(3)" &)
sleep 1
python -c "
from nanollama_amd import _lib
print('B count while A alive', _lib.lib().nl_device_count())"
sleep 3
env | grep -i -E "VISIBLE|ROCR|HIP_|HSA" | head
echo "--- under torchrun 2 procs"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/_cnt.py 2>&1 | grep -E "count|VISIBLE" | head

#!/bin/bash
# Two bench ranks on ONE GPU (both LOCAL_RANK=0): RCCL refuses the duplicate device, the ranks fall back to the socket
# communicator and each computes the factors - exercises the multi-rank bench path end to end on a 1-GPU box.
cd $GRAFT_REPO_ROOT
PORT=29611
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout 300 python bench.py --gpus 2 --size 8192 --batch 16 \
      --steps 5 --warmup 2 > gpurun_out/two_ranks_$r.out 2> gpurun_out/two_ranks_$r.err &
done
wait
echo "rank0 stdout lines: $(wc -l < gpurun_out/two_ranks_0.out), rank1 stdout lines: $(wc -l < gpurun_out/two_ranks_1.out)"
python3 -c "
import json; d=json.loads(open('gpurun_out/two_ranks_0.out').read())
print(d['value'], d['n_gpus'], d['config']['instances_total'], d['config']['collectives'])"
grep -h "bench.py:" gpurun_out/two_ranks_*.err | head -4

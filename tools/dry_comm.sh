cd $GRAFT_REPO_ROOT
export TRAJOPT_WORKER_SAME_DEVICE=1 NCCL_DEBUG=WARN
rm -f /tmp/id.bin
timeout 120 python tests/multi_gpu_worker.py 0 2 12 /tmp/id.bin /tmp/gather > gpurun_out/dry_rank0.log 2>&1 &
timeout 120 python tests/multi_gpu_worker.py 1 2 12 /tmp/id.bin /tmp/gather > gpurun_out/dry_rank1.log 2>&1 &
wait
echo rank0; tail -12 gpurun_out/dry_rank0.log; echo rank1; tail -12 gpurun_out/dry_rank1.log

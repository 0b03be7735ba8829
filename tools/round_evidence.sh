cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_final_r06.log 2>&1
timeout 900 bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err
SIFTMI_RCCL_LIB=$GRAFT_REPO_ROOT/tests/c/libfake_rccl.so timeout 600 python bench.py --gpus 8 --share-gpu --frames 8 --batch 8 --steps 10 --warmup 2 --no-cpu --no-extras --no-roofline > gpurun_out/bench_eight_ranks_share_gpu_r06.json 2> gpurun_out/bench_eight_ranks_share_gpu_r06.err
timeout 400 bash tools/pmc_pipeline.sh "descriptor_kernel" 8 8 dense > gpurun_out/pmc_descriptor_dense_r06_final.txt 2>&1
timeout 400 bash tools/pmc_pipeline.sh "orientation_kernel" 8 8 dense > gpurun_out/pmc_orientation_dense_r06_final.txt 2>&1
timeout 1500 python tools/fuzz_parity.py 120 666006 nspo > gpurun_out/fuzz_parity_r06_final.log 2>&1
timeout 300 python tools/desc_margin.py dense > gpurun_out/desc_margin_final_r06.log 2>&1
tail -n 2 gpurun_out/pytest_final_r06.log; tail -n 1 gpurun_out/fuzz_parity_r06_final.log

python -m pytest tests/test_conv2d_gpu.py tests/test_model_gpu.py tests/test_inputdata_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e2e_trace -o t -- python3 $GRAFT_REPO_ROOT/tools/e2e_profile.py 8 > $GRAFT_REPO_ROOT/gpurun_out/e2e_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_one_forward.py gpurun_out/e2e_trace 60 > gpurun_out/e2e_one_forward.txt
rm -rf gpurun_out/e2e_trace
cat gpurun_out/e2e_one_forward.txt

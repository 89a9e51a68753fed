#!/bin/bash
# One GPU-box session: the steps are the arguments, the outputs go to gpurun_out/${TAG}_<step>.*
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'TAG=r03 bash tools/gpu_session.sh p2p bench1 benchhost gpu'
# steps: p2p | hosttests | rtc | gpu (the whole -m gpu suite) | smoke | bench1 (python bench.py) | benchdriver (the driver's
#        command) | benchhost (--config host) | bench2p2p / bench2de / bench2gloo (two processes sharing the GPU) | trace2 |
#        ipclegacy (the hipIpc test with HSA_ENABLE_IPC_MODE_LEGACY=1) | soakp2p (3000 updates over the peer-to-peer transport, 2 and 4 processes)
TAG=${TAG:-r03}
set -x
for step in "$@"; do
  case $step in
    p2p) timeout -k 10 600 python -m pytest tests/test_p2p.py -q -m gpu > gpurun_out/${TAG}_p2p.log 2>&1; echo "p2p rc=$?"; tail -3 gpurun_out/${TAG}_p2p.log ;;
    bench1) timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench1.json 2> gpurun_out/${TAG}_bench1.err; echo "bench1 rc=$?" ;;
    bench2p2p) timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --p2p on > gpurun_out/${TAG}_bench2_p2p.json 2> gpurun_out/${TAG}_bench2_p2p.err; echo "bench2 p2p rc=$?" ;;
    bench2gloo) timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --p2p off --no-cpu-baseline --repeats 2 > gpurun_out/${TAG}_bench2_gloo.json 2> gpurun_out/${TAG}_bench2_gloo.err; echo "bench2 gloo rc=$?" ;;
    bench2de) timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --p2p on --proposal de > gpurun_out/${TAG}_bench2_p2p_de.json 2> gpurun_out/${TAG}_bench2_p2p_de.err; echo "bench2 de rc=$?" ;;
    ipclegacy) HSA_ENABLE_IPC_MODE_LEGACY=1 timeout -k 10 200 python -m pytest tests/test_p2p.py -q -m gpu -k "over_hip_ipc and rw" > gpurun_out/${TAG}_ipc_legacy1.log 2>&1; echo "ipc legacy=1 rc=$?"; tail -5 gpurun_out/${TAG}_ipc_legacy1.log ;;
    trace2) SABC_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 2 --dist-backend gloo --p2p on --no-cpu-baseline --repeats 2 --steps 20 > gpurun_out/${TAG}_trace2.json 2> gpurun_out/${TAG}_trace2.err; echo "trace2 rc=$?"; grep -v amdgpu.ids gpurun_out/${TAG}_trace2.err | tail -30 ;;
    hosttests) timeout -k 10 600 python -m pytest tests/test_gpu_host_fdist.py tests/test_host_prior.py -q -m gpu -x > gpurun_out/${TAG}_hosttests.log 2>&1; echo "hosttests rc=$?"; tail -5 gpurun_out/${TAG}_hosttests.log ;;
    benchhost) timeout -k 10 300 python bench.py --config host > gpurun_out/${TAG}_bench_host.json 2> gpurun_out/${TAG}_bench_host.err; echo "benchhost rc=$?"; tail -3 gpurun_out/${TAG}_bench_host.err ;;
    rtc) timeout -k 10 900 python -m pytest tests/test_user_simulator.py -q -m gpu -x > gpurun_out/${TAG}_rtc.log 2>&1; echo "rtc rc=$?"; tail -15 gpurun_out/${TAG}_rtc.log ;;
    benchdriver) timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err; echo "benchdriver rc=$?" ;;
    smoke) timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log ;;
    soakp2p) for spec in "2 randomwalk" "2 de" "4 randomwalk" "4 stretch"; do set -- $spec; timeout -k 10 300 python bench.py --gpus $1 --dist-backend gloo --p2p on --proposal $2 --steps 3000 --warmup 5 --repeats 1 --no-cpu-baseline > gpurun_out/${TAG}_soak_p2p_$1_$2.json 2> gpurun_out/${TAG}_soak_p2p_$1_$2.err; echo "soak $1 $2 rc=$?"; done ;;
    gpu) timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "gpu rc=$?"; tail -5 gpurun_out/${TAG}_gpu_tests.log ;;
  esac
done

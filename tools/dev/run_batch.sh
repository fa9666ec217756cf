R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
gcc -std=c99 -O0 -g -Iinclude examples/capi_vcycle.c -o /tmp/capi_vcycle -Lhomogenization.jl_amd -lhmg_hip -Wl,-rpath,$R/homogenization.jl_amd
stdbuf -o0 -e0 /tmp/capi_vcycle 4 4 4; echo "rc=$?"
timeout -k 10 200 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args /tmp/capi_vcycle 4 4 4 2>&1 | tail -40
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t10.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t10.log

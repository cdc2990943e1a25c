mkdir -p gpurun_out/wide
timeout -k 10 400 python tools/ab_libs.py pathtracer_amd/ab/libptcore_base.so pathtracer_amd/ab/libptcore_wide.so pathtracer_amd/ab/libptcore_wide2.so > gpurun_out/wide/ab.log 2>&1 || { tail gpurun_out/wide/ab.log; exit 1; }
grep median gpurun_out/wide/ab.log
timeout -k 10 400 python tools/ab_libs.py --bvh device pathtracer_amd/ab/libptcore_base.so pathtracer_amd/ab/libptcore_wide.so pathtracer_amd/ab/libptcore_wide2.so > gpurun_out/wide/ab_dev.log 2>&1 || { tail gpurun_out/wide/ab_dev.log; exit 1; }
grep median gpurun_out/wide/ab_dev.log
cp pathtracer_amd/ab/libptcore_wide2.so pathtracer_amd/libptcore.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread -k "not wavefront" > gpurun_out/wide/tests.log 2>&1; tail -15 gpurun_out/wide/tests.log

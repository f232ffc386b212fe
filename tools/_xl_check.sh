mkdir -p gpurun_out
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 600 python tools/gpu_time.py kb_f1_N50 0 256 > gpurun_out/f1_phases.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 600 python tools/gpu_time.py merge6_N25 0 256 > gpurun_out/m6_phases.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 600 python tools/gpu_time.py kb_barc3_N25 0 512 > gpurun_out/b3_phases.txt 2>&1
grep -E "scen/s|jacobi|qp  |corr" gpurun_out/f1_phases.txt gpurun_out/m6_phases.txt gpurun_out/b3_phases.txt

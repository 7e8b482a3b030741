# raw outputs of the round-3 kernel tools -> gpurun_out/r3raw/ (copied to profiles/r03_raw/)
O=gpurun_out/r3raw; mkdir -p $O
timeout 600 python tools/wino4_check.py 2>&1 | grep -v amdgpu.ids > $O/wino4_check.txt
timeout 600 python tools/wino4_policy.py 2>&1 | grep -v amdgpu.ids > $O/wino4_policy.txt
timeout 300 python tools/wino4_ksweep.py 2>&1 | grep -v amdgpu.ids > $O/wino4_ksweep.txt
timeout 300 python tools/wino4_time.py 2>&1 | grep -v amdgpu.ids > $O/wino4_time.txt
for w in "sngan32" "sngan64" "sngan32 --phase 2"; do python tools/host_calls.py $w 2>&1 | grep -v "amdgpu.ids\|Fixed Random" | head -12; done > $O/host_calls.txt
for c in 0 2; do for w in "sngan32" "sngan64" "sngan32 --phase 2"; do python tools/host_time.py $w --cores $c 2>/dev/null | grep HOST_TIME; done; done > $O/host_time.txt
tail -3 $O/wino4_check.txt

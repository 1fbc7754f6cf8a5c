for rnd in 1 2 3; do
  for which in old new; do
    cp gpurun_ab/lib_$which.so gomatching_amd/libgomatching_hip.so
    timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-4s %8.2f frames/s %7.3f ms | k256 %.1f us  ffn %.1f  proj_ln %.1f  msda %.1f  bneck %.1f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline_k256_long']['avg_launch_us'], d['roofline_fused_ffn']['avg_launch_us'], d['roofline_proj_ln']['avg_launch_us'], d['roofline_msda']['avg_launch_us'], d['roofline_bneck']['avg_launch_us']))" $which
  done
done

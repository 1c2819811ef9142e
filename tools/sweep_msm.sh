for p in "msm.log_seg=8" "msm.log_seg=6" "msm.log_seg=5" "msm.window_bits=15" "msm.window_bits=14" "msm.window_bits=15,msm.log_seg=6" "msm.log_red_chunk=2" "msm.log_red_chunk=4"; do
  echo "== $p"; SG_PARAMS=$p python bench.py --steps 5 --warmup 1 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['msm_phases_ms'].items()})"
done

for a in 0 2 4 6; do
  VSTAB_ABLATE=$a VSTAB_LIB_PATH=video-annotator_amd/lib/libvstab_dev.so timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate $a', d['value'], 'fps; warp in pipeline', d['roofline']['avg_launch_us'], 'us, alone', d['roofline']['alone']['avg_launch_us'], d['parity_check'])"
done

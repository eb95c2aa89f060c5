for c in 0 1; do for b in 262144 1048576 131072; do
echo -n "CODES=$c batch=$b: "; PCX_WM_CODES=$c python bench.py --game warehouse --batch $b --no-cpu-baseline --no-other-configs --steps 100 --warmup 60 --repeats 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['repeats']['kernel_ms_all'], d['roofline']['frac'], d['roofline']['launch_shape'])"
done; done

out=gpurun_out/r06lens; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=600 -x -k "mid_length or long_reads" 2>&1 | tail -2
LENS="300 500 700 1000 2000" bash tools/len_sweep.sh $out/sweep
LENS="700 1000 2000" CRASS_NO_DENSE_LIGHT=1 bash tools/len_sweep.sh $out/sweep_nolight
for L in 300 1000; do n=$((1500000000 / L)); for p in w=7 d=20,D=40; do for e in CRASS_X=1 CRASS_NO_DENSE_LIGHT=1; do
    env $e python bench.py --read-len $L --total-reads $n --params $p --steps 5 --warmup 2 --cpu-sample 20000 --single-shots 0 --e2e-reads 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('L=$L $p $e', 'ms/step', d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['roofline']['per_kernel'].items()}, 'parity', (d.get('parity_checked') or {}).get('equal'))"
done; done; done

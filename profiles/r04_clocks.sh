# what clock and power the chip holds under the bench step (rocm-smi samples while 3000 steps run)
mkdir -p gpurun_out/r4j
( for i in $(seq 1 90); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.5; done ) > gpurun_out/r4j/smi.txt &
SP=$!
python3 bench.py --steps 3000 --warmup 20 --no-cpu-baseline > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.log
kill $SP 2>/dev/null
tail -1 gpurun_out/r4j/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
grep -n "timed region" gpurun_out/r4j/bench.log
cat gpurun_out/r4j/smi.txt

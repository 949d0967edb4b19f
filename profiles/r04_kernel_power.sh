# clock and power each hot kernel holds when it runs alone, back to back: rocm-smi sampled every 0.25 s while long microbench
# loops run; the samples inside each loop's window with power above 500 W are averaged
mkdir -p gpurun_out/r4k
python3 -c "import torch; torch.zeros(1).cuda()"        # page the image in before anything is timed
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr -d '()Mhz' | tr '\n' ' ')"; sleep 0.2; done ) > gpurun_out/r4k/smi.txt &
SP=$!
for t in mlp384:20000 mlp192:25000 mlp96:25000 dwconv96:60000 dwconv192:90000 dwconv384:150000 pw1_768:60000 pw2_768:60000; do
  n=${t%%:*}; it=${t##*:}
  echo "BEGIN $n $(date +%s.%N)" >> gpurun_out/r4k/windows.txt
  python3 profiles/microbench.py $n $it 2>/dev/null | grep -v amdgpu > gpurun_out/r4k/mb_$n.txt
  echo "END $n $(date +%s.%N)" >> gpurun_out/r4k/windows.txt
  echo "$n: $(tail -1 gpurun_out/r4k/mb_$n.txt)"
done
kill $SP
python3 - <<'PY'
import re
sm=[l.split() for l in open('gpurun_out/r4k/smi.txt') if len(l.split())>=3]
sm=[(float(a),float(b),float(c)) for a,b,c in (x[:3] for x in sm)]
w={}
for l in open('gpurun_out/r4k/windows.txt'):
    k,n,t=l.split(); w.setdefault(n,{})[k]=float(t)
for n,d in w.items():
    xs=[(c,p) for t,c,p in sm if d['BEGIN']<t<d['END'] and p>500]
    if xs: print(f"{n:10s} {len(xs):3d} samples above 500 W: sclk {sum(c for c,_ in xs)/len(xs):6.0f} MHz (min {min(c for c,_ in xs):.0f}, max {max(c for c,_ in xs):.0f})  power {sum(p for _,p in xs)/len(xs):6.0f} W (max {max(p for _,p in xs):.0f})")
    else: print(n, "no loaded samples")
PY

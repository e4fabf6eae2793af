#!/bin/bash
# round 4: decomposition of k_cprod<2> (tools/ubench/cprod_parts.hip) with the shader clock sampled while each variant runs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04parts; mkdir -p $O
(while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Average Graphics Package Power|Current Socket Graphics Package Power' | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*: //' | tr '\n' ' ')"; done) > $O/sclk.txt &
SMI=$!
./tools/ubench/cprod_parts 60 2>&1 | while read l; do echo "$(date +%s.%N | cut -c1-14) $l"; done | tee $O/parts.txt
kill $SMI
python3 - <<P
rows=[l.split() for l in open('$O/sclk.txt') if len(l.split())>=2]
ev=[(float(l.split()[0]), ' '.join(l.split()[1:])) for l in open('$O/parts.txt')]
prev=None
for t,name in ev:
    if prev is not None:
        s=[r for r in rows if prev < float(r[0]) < t]
        clk=[int(r[1].replace('Mhz','')) for r in s if r[1].endswith('Mhz')]
        pw=[float(r[2]) for r in s if len(r)>2 and r[2].replace('.','',1).isdigit()]
        print(name[:110], '| sclk avg', round(sum(clk)/max(1,len(clk))), 'MHz over', len(clk), 'samples', '| power', round(sum(pw)/max(1,len(pw))), 'W')
    prev=t
P

import json,sys
d=json.load(open(sys.argv[1]))
for br in ('lr','hr'):
    L=d[br]; tot=0
    print(br)
    for r in sorted(L,key=lambda r:-r['us_per_call']*r['calls']):
        t=r['us_per_call']*r['calls']; tot+=t
        if 'wino' in str(r['plan']) or len(sys.argv)>2:
            print(f"{r['N']:3d} {r['H']:4d}x{r['W']:<4d} {r['cin']:4d}->{r['cout']:<4d} k{r['k']} d{r['dil']} up{int(r['up2'])} {str(r['plan']):12s} calls {r['calls']:.0f} us {t:8.1f} frac {r['frac_of_peak']:.3f}", {k:round(v) for k,v in r['us_by_kernel'].items()})
    print(tot)

import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step"]
        print(f.split('/')[-1], d["ms_per_step"], ' '.join('%s=%.3f'%(a.replace('smpf_',''),k[a]) for a in list(k)[:10]))
    except Exception as e: print(f, 'ERR', e)

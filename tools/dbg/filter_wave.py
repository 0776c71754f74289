import json,sys
for l in sys.stdin:
    if l.startswith("=="): print(l.strip()); continue
    try: r=json.loads(l)
    except Exception: continue
    if r.get('kernel','').startswith('k_req_wave'): print("  %-30s %8.3f ms %7.1f GB/s" % (r['case'], r['ms'], r['GBps']))

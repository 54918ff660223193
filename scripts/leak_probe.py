"""Development aid: does a Runner give its device memory back?  Builds the bench's runner at a small size, runs two iterations,
drops it, and reports what is still allocated and who still refers to the engine.   python scripts/leak_probe.py [res,spp ...]"""
import gc, os, sys, weakref
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
gib = lambda: torch.cuda.memory_allocated(dev) / 2 ** 30
free = lambda: "free %.2f / %.2f GiB (driver), reserved by torch %.2f GiB" % (torch.cuda.mem_get_info(dev)[0] / 2 ** 30, torch.cuda.mem_get_info(dev)[1] / 2 ** 30,
                                                                            torch.cuda.memory_reserved(dev) / 2 ** 30)
print("allocated at start: %.2f GiB; %s" % (gib(), free()))
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(224, 64)]
for res, spp in cfgs[:-1]:       # the runners before the last one: built, run, dropped like bench.py's legs
    r = bench.build_runner(res, spp, False, dev)
    bench.timed_steps(r, 2, 2, dev)
    print("%d^2 x %d: allocated with the runner alive: %.2f GiB; %s" % (res, spp, gib(), free()))
    del r
    gc.collect(); torch.cuda.empty_cache()
    print("   after del + gc + empty_cache: allocated %.2f GiB; %s" % (gib(), free()))
res, spp = cfgs[-1]
r = bench.build_runner(res, spp, False, dev)
bench.timed_steps(r, 2, 2, dev)
print("allocated with the runner alive: %.2f GiB; %s; plan %s" % (gib(), free(), r.renderer.engine.plan(res * res, spp)))
eng_ref, run_ref = weakref.ref(r.renderer.engine), weakref.ref(r)
del r
gc.collect(); torch.cuda.empty_cache()
print("allocated after del + gc: %.2f GiB; runner alive: %s, engine alive: %s" % (gib(), run_ref() is not None, eng_ref() is not None))
for name, ref in (("runner", run_ref), ("engine", eng_ref)):
    o = ref()
    if o is None:
        continue
    for ref_by in gc.get_referrers(o):
        if ref_by is locals() or ref_by is globals():
            continue
        desc = type(ref_by).__name__
        if isinstance(ref_by, dict):
            owners = [type(x).__name__ + ":" + getattr(type(x), "__module__", "") for x in gc.get_referrers(ref_by)][:4]
            desc += " keys=%s owned by %s" % (list(ref_by.keys())[:8], owners)
        print("  %s is referred to by %s" % (name, desc[:300]))
    del o
if os.environ.get("LEAK_SNAPSHOT"):
    for seg in torch.cuda.memory_snapshot():
        if seg["total_size"] > (1 << 30):
            live = [(b["size"], b["state"]) for b in seg["blocks"] if b["state"] != "inactive"]
            print("segment %.2f GiB, stream %s, pool %s: %d blocks, live: %s" % (seg["total_size"] / 2 ** 30, seg.get("stream"), seg.get("segment_pool_id"),
                                                                                  len(seg["blocks"]), live[:8]))

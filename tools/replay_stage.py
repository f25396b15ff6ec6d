import sys, os, json, hashlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
import shard_replay as sr
import torch.multiprocessing as mp
from luminair_amd import backend
def stage(ctx, bufs):
    ctx.set_profiling(True); ctx.prove_tables(bufs); ctx.set_profiling(False)
    t = ctx.timings(); return {k: round(v, 3) for k, v in t.items() if k.endswith('_ms')}
if __name__ == '__main__':
    name, world = sys.argv[1], int(sys.argv[2])
    ctx = backend.Context(0); tabs = sr.workload(name)
    bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
    ctx.prove_tables(bufs); print('unsharded', json.dumps(stage(ctx, bufs)))
    for _, b, _ in bufs: b.free()
    ctx.close()
    path = '/tmp/rec.npz'; mpc = mp.get_context('spawn'); q = mpc.Queue(); port = sr._free_port()
    procs = [mpc.Process(target=sr.record_worker, args=(r, world, port, name, path, q)) for r in range(world)]
    [p.start() for p in procs]; [q.get(timeout=1200) for _ in range(world)]; [p.join() for p in procs]
    data = np.load(path); rec = [data['arr_%d' % i] for i in range(len(data.files))]
    ctx = backend.Context(0); staged = [ctx.upload(r) for r in rec]; st = {'i': 0}
    def ag(buf, nbytes, _s):
        k = st['i'] % len(rec); st['i'] += 1; ctx.device_copy(buf, staged[k].ptr, nbytes * world)
    ctx.set_shard(0, world, ag)
    bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
    ctx.prove_tables(bufs); print('rank0 of %d' % world, json.dumps(stage(ctx, bufs)))

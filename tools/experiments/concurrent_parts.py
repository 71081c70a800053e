# two (or K) concurrent half-renders on one GPU vs one whole render: is there idle capacity a pipelined pass could use?
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bhraytracer_amd as B
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
scene_rel, W, H, spp, gi = bench.WORKLOADS[wl]
path = os.path.join(bench.ROOT, scene_rel)
def mk(rank, world):
    sc = B.Scene(path); sc.upload(0)
    o = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0, rank=rank, world_size=world, tile_size=32)
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    return sc, o, rgb, rad
one = mk(0, 1)
def run(x): x[0].render_dev(x[1], x[2].data_ptr(), x[3].data_ptr())
run(one); run(one); torch.cuda.synchronize()
t0 = time.perf_counter(); 
for _ in range(3): run(one)
torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / 3
one[0].close(); del one; torch.cuda.empty_cache()
parts = [mk(r, K) for r in range(K)]
for p in parts: run(p)
for p in parts: run(p)
torch.cuda.synchronize()
# sequential parts
t0 = time.perf_counter()
for _ in range(3):
    for p in parts: run(p)
torch.cuda.synchronize(); t_seq = (time.perf_counter() - t0) / 3
def loop(p):
    for _ in range(3): run(p)
t0 = time.perf_counter()
th = [threading.Thread(target=loop, args=(p,)) for p in parts]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t_con = (time.perf_counter() - t0) / 3
print(f"{wl}: whole frame {t_one*1e3:.2f} ms; {K} tile-parts one after the other {t_seq*1e3:.2f} ms; {K} parts concurrently {t_con*1e3:.2f} ms")

import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "cornerstone-octree_amd"))
import torch, cstone_amd
ctx = cstone_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(3)
# leave recognisable garbage in LDS / scratch first
big = torch.randint(0, 2**30, (1 << 20,), dtype=torch.int32, device="cuda", generator=g)
b0 = big.clone()
ctx.sort_pairs(big, torch.arange(1 << 20, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
print("big ok", bool((big == torch.sort(b0).values).all()), flush=True)
try:
    ctx.sync(); print("big noerr")
except Exception as e:
    print("big", str(e)[-40:])
for n in (1, 3, 65, 4095, 4097):
    keys = torch.randint(0, 2**30, (n,), dtype=torch.int32, device="cuda", generator=g)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    k0 = keys.clone()
    ctx.sort_pairs(keys, vals)
    torch.cuda.synchronize()
    ref = torch.sort(k0).values
    ok = bool((keys == ref).all())
    try:
        ctx.sync(); err = "noerr"
    except Exception as e:
        err = str(e)[-120:]
    print(n, ok, err, keys[:4].tolist(), ref[:4].tolist(), flush=True)

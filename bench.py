#!/usr/bin/env python3
"""bench.py -- particles/sec through domain.sync (encode + sort + tree + halo) on MI355X.

One "step" = one steady-state `sync` of the hot path over the resident particle set, in a time-stepping loop: before
every sync EVERY particle is displaced by up to 0.1 h per coordinate (the displacement is the client's work: it runs
outside the timed intervals, every sync is bracketed on its own); (SURVEY.md section 3.1,
reference include/cstone/domain/domain.hpp:196-243 with the stages a rank executes):
    bounding box (min/max of x,y,z) -> SFC key encode -> stable radix sort of (key, index) ->
    global-tree rebalance step + node counts -> gather h -> focus-tree rebalance step + counts ->
    linked octree + geometric centers -> halo radii (segment max of h) + halo discovery traversal ->
    layout scan -> gather x,y,z into SFC order.
Inputs are synthetic uniform-random particles, resident in HBM before the timed region starts.

Usage: python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts N ranks itself -- a child
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, one rank per GPU over
  RCCL -- BEFORE it makes any GPU call, relays rank 0's JSON line and exits with the child's code.  Under a launcher
  (WORLD_SIZE / RANK / LOCAL_RANK set) it is one of the ranks.
  --path single | mr   which Domain::sync runs: cstone_hip_domain_sync_scratch (one rank only) or the multi-rank
                       cstone_hip_domain_mr_sync (default for N > 1; with N = 1 an RCCL world of one rank)
  --dist uniform | plummer | clustered   the cloud (cstone_amd/clouds.py)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--particles", type=float, default=1e8, help="GLOBAL particle count (strong scaling over --gpus)")
    p.add_argument("--key-bits", type=int, default=64)
    p.add_argument("--real-bits", type=int, default=64)
    p.add_argument("--curve", default="hilbert", choices=["hilbert", "morton"])
    p.add_argument("--bucket-focus", type=int, default=64)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-plummer", action="store_true", help="skip extras.plummer (BASELINE configs[2])")
    p.add_argument("--no-variants", action="store_true", help="skip the sort / movement variants of extras (tuning runs)")
    p.add_argument("--no-stage-timers", action="store_true",
                   help="tuning runs: no HIP events around the launches (no roofline, no stage times): what do they cost?")
    p.add_argument("--neighbor-targets", type=float, default=1e7,
                   help="after the timed region: findNeighbors for this many particles of the synced domain (0: skip)")
    p.add_argument("--cpu-sample", type=float, default=4e6, help="particles in the CPU baseline sample")
    p.add_argument("--path", default=None, choices=["single", "mr"],
                   help="single: cstone_hip_domain_sync_scratch (N = 1 only); mr: cstone_hip_domain_mr_sync with RCCL "
                        "collectives (default for N > 1; at N = 1 a world of one rank: the same code path as the N-GPU run)")
    p.add_argument("--dist", default="uniform", choices=["uniform", "plummer", "clustered"],
                   help="the particle cloud (cstone_amd/clouds.py): BASELINE configs[3] / [2] / [4]")
    p.add_argument("--no-mr-extra", action="store_true",
                   help="N = 1, --path single: skip extras.mr_path_world_of_one (the same cloud through the multi-rank path)")
    p.add_argument("--launch-only", action="store_true",
                   help="start the ranks, let each print its RANK / WORLD_SIZE / LOCAL_RANK and leave (no GPU work): the "
                        "check that --gpus N really starts N ranks")
    p.add_argument("--master-port", type=int, default=0, help="rendezvous port of the ranks this process starts (0: a free one)")
    return p.parse_args()


def launch_ranks(args):
    """--gpus N > 1 and no launcher around us: start the N ranks (one per GPU, torch.distributed.run over 127.0.0.1) as a
    CHILD of this process, which has not touched the GPU and never will; relay what rank 0 prints; return the child's
    exit code.  (The reference starts its GPU ranks with one MPI rank per device the same way,
    test/integration_mpi/CMakeLists.txt:53-59.)"""
    import socket
    import subprocess

    port = args.master_port
    if not port:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CSTONE_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:  # rank 0's JSON line (and, with --launch-only, one line per rank)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


class SyncPipeline:
    """Steady-state domain.sync through the C ABI: cstone_hip_domain_sync on device-resident arrays."""

    def __init__(self, ctx, n, key_bits, real_bits, curve, bucket, bucket_focus, seed, dist="uniform"):
        import torch

        import cstone_amd
        from cstone_amd import clouds
        from cstone_amd.domain import Domain

        self.ctx, self.n, self.kb, self.rb, self.dist = ctx, n, key_bits, real_bits, dist
        cv = cstone_amd.HILBERT if curve == "hilbert" else cstone_amd.MORTON
        dev = ctx.device
        rdt = torch.float64 if real_bits == 64 else torch.float32
        # the cloud (cstone_amd/clouds.py): uniform in the unit cube, the reference's Plummer sphere
        # (test/coord_samples/plummer.hpp restated, srand48(42)) or the 8-blob mixture; open boundaries: the domain
        # measures its box itself (makeGlobalBox + limitBoxShrinking), `lim` is only where it starts from
        self.x, self.y, self.z, self.h, lim = clouds.make_cloud(dist, n, n, dev, rdt, seed)
        self.keys = torch.zeros(n, dtype=cstone_amd.key_torch_dtype(key_bits), device=dev)
        # the scratch tuple of the client (R/domain/domain.hpp:196-206: at least three vectors there as well)
        self.scratch = [torch.empty(n, dtype=rdt, device=dev) for _ in range(int(os.environ.get("CSTONE_BENCH_SCRATCH", "4")))]
        self.dom = Domain(ctx, cv, key_bits, real_bits, bucket, bucket_focus, 0.5, cstone_amd.make_cbox(lim))
        self.f_leaves = self.g_leaves = 0

    def step(self):
        """one Domain::sync (domain.hpp:196-243); the first call also converges both trees from the root.  Straight
        through the C ABI, as a C++ client would call it: the particle arrays and a scratch tuple of three buffers (like
        the reference's sync takes) that the library may exchange among themselves; the tensors follow their buffers (no
        particle is removed here, so every array keeps its length)"""
        import ctypes as C

        arrays = (self.x, self.y, self.z, self.h, *self.scratch)
        ns = len(self.scratch)
        if not hasattr(self, "_ptrs"):
            self._keys_ptr = C.c_void_p(self.keys.data_ptr())
            self._ptrs = [C.c_void_p() for _ in range(4)]
            self._sarr = (C.c_void_p * ns)()
            self._n = C.c_size_t(self.x.numel())
            self._none = C.c_void_p(None)
        by_ptr = {}
        for p, t in zip(self._ptrs, arrays[:4]):
            p.value = t.data_ptr()
            by_ptr[p.value] = t
        for q, t in enumerate(self.scratch):
            self._sarr[q] = t.data_ptr()
            by_ptr[t.data_ptr()] = t
        rc = self.ctx.lib.cstone_hip_domain_sync_scratch(self.dom.h, C.byref(self._keys_ptr), C.byref(self._ptrs[0]),
                                                         C.byref(self._ptrs[1]), C.byref(self._ptrs[2]),
                                                         C.byref(self._ptrs[3]), self._n, self._sarr, C.c_int(ns),
                                                         self._none, self._none, C.c_int(0))
        self.ctx._chk(rc, "domain_sync")
        self.x, self.y, self.z, self.h = [by_ptr[p.value] for p in self._ptrs]
        self.scratch = [by_ptr[self._sarr[q]] for q in range(ns)]

    def first_sync(self):
        self.step()
        self.note_leaves()

    def note_leaves(self):
        v = self.dom.view()
        assert v.num_particles_with_halos == self.x.numel()
        self.f_leaves, self.g_leaves = v.num_focus_leaves, v.num_global_leaves

    def jiggle(self):
        """displace a random 1 % of the particles by up to 2h (what a time step of a simulation does to the order)"""
        import torch

        if not hasattr(self, "g"):
            self.g = torch.Generator(device=self.x.device).manual_seed(1234)
        n = self.x.numel()
        m = max(1, n // 100)
        idx = torch.randint(0, n, (m,), device=self.x.device, generator=self.g)
        hh = self.h[idx]
        for a in (self.x, self.y, self.z):
            d = (torch.rand(m, dtype=a.dtype, device=a.device, generator=self.g) - 0.5) * (4 * hh)
            a[idx] = (a[idx] + d).clamp_(0.0, 1.0) if self.dist == "uniform" else a[idx] + d

    def drift(self, frac=0.1, clamp=None):
        """displace EVERY particle by up to frac * h per coordinate (a time step at a Courant number of that order).  The
        uniform cloud of the headline is kept inside [0, 1] (its box then stays what it is: the steady state of a
        periodic or walled simulation); a Plummer sphere or the blobs are NOT clamped: their outermost particles move and
        the open box follows them"""
        import torch

        if not hasattr(self, "g"):
            self.g = torch.Generator(device=self.x.device).manual_seed(1234)
        clamp = (self.dist == "uniform") if clamp is None else clamp
        for a in (self.x, self.y, self.z):
            d = torch.rand(a.numel(), dtype=a.dtype, device=a.device, generator=self.g)
            a.add_(d.sub_(0.5).mul_(2 * frac).mul_(self.h))
            if clamp:
                a.clamp_(0.0, 1.0)
            del d

    def drift_unclamped(self, frac=0.1):
        """the same displacement without keeping the particles inside [0, 1]: the outermost particles of the open box
        move, so the box of the reference's limitBoxShrinking rule changes with every sync"""
        self.drift(frac, clamp=False)

    def find_neighbors(self, targets, ngmax):
        """cstone_hip_find_neighbors on the synced domain's own tree view (NOT part of the timed metric)"""
        import ctypes as C

        import torch

        v = self.dom.view()
        n = v.end_index - v.start_index
        nt = min(int(targets), n)
        first = v.start_index + (n - nt) // 2
        counts = torch.zeros(nt, dtype=torch.int32, device=self.x.device)
        nidx = torch.zeros(max(1, nt * ngmax), dtype=torch.int32, device=self.x.device)
        ctx = self.ctx
        ctx.profile_reset()
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        rc = ctx.lib.cstone_hip_find_neighbors(ctx.h, C.c_int(self.rb), P(self.x), P(self.y), P(self.z), P(self.h),
                                               C.c_uint32(first), C.c_uint32(first + nt), C.byref(v.box),
                                               C.c_void_p(v.child_offsets), C.c_void_p(v.internal_to_leaf),
                                               C.c_void_p(v.layout), C.c_void_p(v.centers), C.c_void_p(v.sizes),
                                               C.c_float(1.0), C.c_uint32(ngmax), P(nidx) if ngmax else None,
                                               P(counts))
        ctx._chk(rc, "find_neighbors")
        ctx.sync()
        ms, _ = ctx.profile_get("neighbors")
        out = {"targets": nt, "ngmax": ngmax, "ms": ms, "targets_per_s": nt / (ms * 1e-3) if ms > 0 else None,
               "mean_neighbors": float(counts.double().mean().item()), "max_neighbors": int(counts.max().item())}
        if ngmax == 0:
            # the traversal counters of the reference's NcStats (find_neighbors.cuh:345-369) from one more, untimed run of
            # the instrumented kernel; rates on the time of the plain kernel above.  11 flop per distance test is the
            # reference's own accounting (test/performance/neighbor_driver.cu:169)
            st = (C.c_uint64 * 4)()
            rc = ctx.lib.cstone_hip_find_neighbors_stats(ctx.h, C.c_int(self.rb), P(self.x), P(self.y), P(self.z),
                                                         P(self.h), C.c_uint32(first), C.c_uint32(first + nt),
                                                         C.byref(v.box), C.c_void_p(v.child_offsets),
                                                         C.c_void_p(v.internal_to_leaf), C.c_void_p(v.layout),
                                                         C.c_void_p(v.centers), C.c_void_p(v.sizes), C.c_float(1.0),
                                                         C.c_uint32(0), None, P(counts), st)
            ctx._chk(rc, "find_neighbors_stats")
            sec = ms * 1e-3
            out.update({"p2p_tests": int(st[0]), "p2p_tests_per_target": st[0] / nt, "max_p2p_tests_of_a_target": int(st[1]),
                        "max_stack": int(st[2]), "p2p_tests_issued": int(st[3]),
                        "lane_efficiency": st[0] / st[3] if st[3] else None,
                        "p2p_tests_per_s": st[0] / sec if sec > 0 else None,
                        "gflops_11_per_test": 11.0 * st[0] / sec / 1e9 if sec > 0 else None,
                        "gflops_11_per_issued_test": 11.0 * st[3] / sec / 1e9 if sec > 0 else None})
        return out


class DistributedPipeline:
    """N ranks, one per GPU: cstone_hip_domain_mr_sync -- the multi-rank Domain::sync inside libcstone_hip (global box /
    global tree all-reduce, SFC assignment, particle all_to_all + merge, owner-side halo discovery + halo all_to_all;
    SURVEY.md section 8e) with the collectives served by torch.distributed (RCCL).  Every rank starts with a random 1/N
    of the cloud (the first sync moves (N-1)/N of it); before every step a random 1% of the assigned particles is
    displaced by up to 2h so that the steady-state exchange really moves particles across the boundaries."""

    def __init__(self, ctx, n_local, n_global, key_bits, real_bits, curve, bucket, bucket_focus, seed, dist="uniform",
                 rank=0, world=1, backend="nccl"):
        import torch

        import cstone_amd
        from cstone_amd import clouds
        from cstone_amd.distributed import NativeDistributedDomain, RcclCollectives

        self.torch, self.dist = torch, dist
        dev = ctx.device
        rdt = torch.float64 if real_bits == 64 else torch.float32
        self.g = torch.Generator(device=dev).manual_seed(seed + 1000 * rank)
        # this rank's share: a random 1/N of the global cloud
        self.x, self.y, self.z, self.h, lim = clouds.make_cloud(dist, n_local, n_global, dev, rdt, seed, rank, world)
        cv = cstone_amd.HILBERT if curve == "hilbert" else cstone_amd.MORTON
        self.native = True  # the multi-rank sync inside libcstone_hip (cstone_hip_domain_mr_sync)
        self.transport = "torch.distributed callbacks"
        coll = None
        self.rccl_ranks = 0
        if backend == "nccl" and os.environ.get("CSTONE_BENCH_TORCH_COLL") != "1":
            # the data path: RCCL from C++ on the library's stream (csrc/comm_rccl.hip); torch.distributed only carries the
            # RCCL id at start-up and serves the bench's own barriers and the max over the ranks
            coll = RcclCollectives(ctx)
            self.rccl_ranks = coll.size  # size of the communicator cstone_hip_comm_rccl_create really built
            self.transport = "RCCL inside libcstone_hip (cstone_hip_comm_rccl, collectives on the library's stream)"
        self.dom = NativeDistributedDomain(ctx, cv, key_bits, real_bits, bucket, bucket_focus, lim, (0, 0, 0), coll=coll)
        self.f_leaves = self.g_leaves = 0
        self.assigned = n_local
        self.halos = 0
        self.stats = {}

    def jiggle(self):
        torch = self.torch
        n = self.x.numel()
        m = max(1, n // 100)
        idx = torch.randint(0, n, (m,), device=self.x.device, generator=self.g)
        hh = self.h[idx]
        for a in (self.x, self.y, self.z):
            d = (torch.rand(m, dtype=a.dtype, device=a.device, generator=self.g) - 0.5) * (4 * hh)
            a[idx] = (a[idx] + d).clamp_(0.0, 1.0) if self.dist == "uniform" else a[idx] + d

    def drift(self, frac=0.1, clamp=None):
        """displace EVERY assigned particle by up to frac * h per coordinate (the same motion as on one GPU)"""
        torch = self.torch
        clamp = (self.dist == "uniform") if clamp is None else clamp
        for a in (self.x, self.y, self.z):
            d = torch.rand(a.numel(), dtype=a.dtype, device=a.device, generator=self.g)
            a.add_(d.sub_(0.5).mul_(2 * frac).mul_(self.h))
            if clamp:
                a.clamp_(0.0, 1.0)
            del d

    def step(self):
        r = self.dom.sync(self.x, self.y, self.z, self.h)
        s, e = r["start"], r["end"]
        # the client owns the assigned particles (updated in place); halos are re-discovered by the next sync
        self.x, self.y, self.z, self.h = r["x"][s:e], r["y"][s:e], r["z"][s:e], r["h"][s:e]
        self.assigned, self.halos = e - s, r["x"].numel() - (e - s)
        self.last = r
        v = self.dom.view()
        self.f_leaves, self.g_leaves = v.num_focus_leaves, v.num_global_leaves
        self.stats = dict(moved=v.particles_sent, halos=v.halos_received, served=v.halos_sent,
                          halo_boxes=v.halo_boxes_exported, peers=v.num_peers)

    first_sync = step

    def invariants(self, n_global):
        """after the timed region: what the reference's multi-rank tests check first (T/integration_mpi/domain_nranks.cpp:
        117-131) -- no particle lost over the ranks, keys sorted, assigned keys inside the rank's own SFC range"""
        import torch.distributed as dist

        torch, r = self.torch, self.last
        s, e = r["start"], r["end"]
        keys = r["keys"]
        ok = bool((keys[1:] >= keys[:-1]).all()) if keys.numel() > 1 else True
        if self.native and e > s:
            v = self.dom.view()
            lo = int(keys[s].item())
            hi = int(keys[e - 1].item())
            ok = ok and lo >= v.range_start and hi < v.range_end
        t = torch.tensor([e - s, 1 if ok else 0], dtype=torch.int64,
                         device=self.x.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t[0].item()) == n_global and int(t[1].item()) == dist.get_world_size()


def granted_cores():
    """the host cores this process may really use: its affinity mask, cut by the cgroup's CPU quota (a lease on a shared
    host grants a share of the machine's cores; os.cpu_count() reports all of them)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_domain(n_sample, bucket, bucket_focus, min_seconds=10.0):
    """the reference's own cstone::Domain<uint64_t,double,CpuTag>::sync on one MPI rank (oracle/_ref, prebuilt)"""
    import ctypes as C

    import numpy as np

    cores = granted_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before the OpenMP runtime of the library starts

    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcstone_ref_domain.so"))
    lib.cstone_refdom_create.restype = C.c_void_p
    d = C.c_void_p(lib.cstone_refdom_create(C.c_uint(bucket), C.c_uint(bucket_focus), C.c_float(0.5),
                                            (C.c_double * 6)(0, 1, 0, 1, 0, 1), (C.c_int * 3)(0, 0, 0)))
    rng = np.random.default_rng(7)
    x, y, z = [rng.uniform(0, 1, n_sample) for _ in range(3)]
    h = np.full(n_sample, 0.01)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lib.cstone_refdom_set(d, C.c_size_t(n_sample), p(x), p(y), p(z), p(h), None)
    lib.cstone_refdom_sync(d)  # first call (tree convergence) is not part of the steady state
    done, t_total = 0, 0.0
    while t_total < min_seconds and done < 60:
        t0 = time.perf_counter()
        lib.cstone_refdom_sync(d)
        t_total += time.perf_counter() - t0
        done += 1
    lib.cstone_refdom_destroy(d)
    return {"value": n_sample * done / t_total, "unit": "particles/s", "cores": cores, "kind": "reference",
            "host_cores_total": os.cpu_count(),
            "sample": f"{done} steady-state Domain<uint64_t,double,CpuTag>::sync of {n_sample} uniform particles on 1 MPI rank, "
                      f"bucket {bucket}, bucketFocus {bucket_focus}, OMP_NUM_THREADS={cores} = the cores granted to this "
                      f"process (the sort inside is a serial std::stable_sort)"}


def cpu_baseline(n_sample, key_bits, real_bits, curve, bucket_focus, min_seconds=8.0):
    """the same stages on the host cores, timed on a bounded sample (reference build if it travelled, else our port)"""
    import numpy as np

    from oracle import oracle as orc

    if key_bits == 64 and real_bits == 64 and curve == "hilbert":
        try:
            return cpu_baseline_domain(n_sample, max(64, n_sample // 100), bucket_focus, min_seconds)
        except OSError:
            pass  # the MPI-linked reference Domain did not travel / load: fall back to the stage-by-stage baseline
    if orc.reference_available():
        impl, kind = orc.Reference(), "reference"
    else:
        if not os.path.exists(orc.Oracle.libpath):
            orc.build("liboracle")
        impl, kind = orc.Oracle(), "port"
    cv = orc.HILBERT if curve == "hilbert" else orc.MORTON
    rng = np.random.default_rng(7)
    rdt = np.float64 if real_bits == 64 else np.float32
    x, y, z = [rng.uniform(0, 1, n_sample).astype(rdt) for _ in range(3)]
    h = np.full(n_sample, 0.01, dtype=rdt)
    tree = counts = None
    done, t_total = 0, 0.0
    while t_total < min_seconds and done < 5:
        t0 = time.perf_counter()
        box = orc.Box([x.min(), x.max(), y.min(), y.max(), z.min(), z.max()])
        keys = impl.compute_sfc_keys(cv, key_bits, x, y, z, box)
        ks, order = impl.sort_pairs(keys, np.arange(n_sample, dtype=np.uint32))
        if tree is None:
            tree, counts = impl.compute_octree(ks, bucket_focus)
        else:
            tree, counts, _ = impl.update_octree(ks, bucket_focus, tree, counts)
        octree = impl.build_octree(tree)
        hs = h[order]
        nl = tree.size - 1
        radii = np.full(nl, 2 * 0.01, dtype=np.float32)
        if cv == orc.HILBERT:
            impl.find_halos(orc.HILBERT, octree, tree, radii, box, 0, nl, real_bits)
        x, y, z, h = x[order], y[order], z[order], hs
        t_total += time.perf_counter() - t0
        done += 1
    return {"value": n_sample * done / t_total, "unit": "particles/s", "cores": impl.num_threads(), "kind": kind,
            "sample": f"{done} sync(s) of {n_sample} uniform particles, stage by stage (encode, sort, tree, linked octree, "
                      f"halo discovery, gathers), bucket {bucket_focus}"}


def main():
    args = parse()
    launched = os.environ.get("WORLD_SIZE") is not None  # under torch.distributed.run (the driver's, or our own child)
    if not launched and args.gpus > 1:
        # N ranks asked for and no launcher around us: start them (before anything here touches the GPU) and relay
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_only:
        line = json.dumps({"launch_only": True, "rank": rank, "world_size": world, "local_rank": local_rank,
                           "gpus_requested": args.gpus,
                           "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}) + "\n"
        os.write(1, line.encode())  # (one write per rank: the ranks share the pipe)
        return
    # the ONE JSON line is all that may appear on stdout: whatever else writes to file descriptor 1 from here on (RCCL
    # prints a version banner there when a communicator is created) goes to stderr, the line itself to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if launched and world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); the world size counts", file=sys.stderr)
    path = args.path or ("mr" if world > 1 or os.environ.get("CSTONE_BENCH_FORCE_DIST") == "1" else "single")
    if path == "single" and world > 1:
        sys.exit("bench.py: --path single is the one-rank Domain::sync; several ranks need --path mr")

    import torch

    import cstone_amd

    # CSTONE_BENCH_BACKEND=gloo: rehearsal of the N-rank bench logic with several processes on ONE GPU (the numbers mean
    # nothing then: host-staged collectives, shared device)
    backend = os.environ.get("CSTONE_BENCH_BACKEND", "nccl")
    distributed = path == "mr"

    def init_group():
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:  # the RCCL world of one rank on one GPU
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if backend != "nccl":
        local_rank = 0
    if distributed:
        init_group()
    torch.cuda.set_device(local_rank)
    ctx = cstone_amd.Context(local_rank)
    if os.environ.get("CSTONE_BENCH_MARKERS") == "1":  # roctx ranges per stage (rocprofv3 --marker-trace)
        ctx.profile_markers(True)

    n_global = int(args.particles)
    n_local = n_global // world
    bucket_global = max(64, n_global // (100 * world))
    if distributed:
        pipe = DistributedPipeline(ctx, n_local, n_global, args.key_bits, args.real_bits, args.curve, bucket_global,
                                   args.bucket_focus, seed=42, dist=args.dist, rank=rank, world=world, backend=backend)
    else:
        pipe = SyncPipeline(ctx, n_local, args.key_bits, args.real_bits, args.curve, bucket_global, args.bucket_focus,
                            seed=42 + rank, dist=args.dist)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_syncs(count, move):
        """count syncs, each bracketed on its own by barrier + device synchronisation on both sides; the particles are
        displaced (by the client: not part of a sync) before each of them; returns the seconds spent inside the syncs"""
        total = 0.0
        for _ in range(count):
            if move:
                move()
            barrier()
            ts = time.perf_counter()
            pipe.step()
            barrier()
            total += time.perf_counter() - ts
        return total

    barrier()
    t_first = time.perf_counter()
    pipe.first_sync()  # converges both trees from the root and moves every particle: reported, not part of `value`
    barrier()
    first_sync_ms = (time.perf_counter() - t_first) * 1e3
    # the headline workload: a time-stepping loop in which EVERY particle is displaced by up to 0.1 h per coordinate
    # between two syncs (about 7 % of them leave their leaf, the trees change a little every time)
    move = pipe.drift
    run_syncs(args.warmup, move)
    # HIP events around the kernels that move the particle arrays only (roofline): every stage of a sync bracketed costs
    # 0.1 ms of it in event records; the full stage table comes from further syncs behind the timed region
    ctx.profile_enable(0 if args.no_stage_timers else 2)
    ctx.profile_reset()
    stats_before = pipe.dom.stats() if not distributed else None
    elapsed = run_syncs(args.steps, move)
    if not distributed:
        pipe.note_leaves()
    n_sorted = n_local
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        n_sorted = pipe.assigned
    timed_stages = {s: ctx.profile_get(s) for s in cstone_amd.STAGES}
    timed_spreads = {s: ctx.profile_spread(s) for s in cstone_amd.STAGES}
    timed_stats = None
    sort_ms_in_syncs = 0.0
    if not distributed:
        after = pipe.dom.stats()
        timed_stats = {k: after[k] - stats_before[k] for k in ("resorts", "resort_fallbacks", "box_redos",
                                                                "full_sort_fallbacks")} | {"last_movers": after["last_movers"]}
    # every stage of a sync, on args.steps further syncs of the same kind (not part of `value`)
    ctx.profile_enable(0 if args.no_stage_timers else 1)
    ctx.profile_reset()
    run_syncs(args.steps, move)
    stage_ms = {s: ctx.profile_get(s)[0] / args.steps for s in cstone_amd.STAGES}
    radix_stats = None
    resorted = timed_stages["resort_leaves"][1] > 0
    if resorted:
        # what is left of the radix pass inside such syncs are the small sorts of the octree build: not the kernel's case
        timed_stages["sort_pass"] = timed_stages["sort_pass_iota"] = (0.0, 0)
    if timed_stages["sort_pass"][1]:
        radix_stats = (timed_stages["sort_pass"][0], timed_stages["sort_pass"][1], timed_spreads["sort_pass"],
                       timed_stages["sort_pass_iota"][0], timed_stages["sort_pass_iota"][1],
                       timed_spreads["sort_pass_iota"], "the digit passes of the timed syncs")
    if distributed:
        # On several ranks a sync also launches the pass kernel on small inputs (the newcomers, the tree's node keys), so
        # the average over ALL launches says nothing about the dominant kernel: it is measured on its own, on as many
        # random pairs as this rank holds, right after the timed region.
        kdt = cstone_amd.key_torch_dtype(args.key_bits)
        rk = torch.randint(0, 1 << (62 if args.key_bits == 64 else 30), (n_sorted,), device=ctx.device).to(kdt)
        rv = torch.empty(n_sorted, dtype=torch.int32, device=ctx.device)
        for rep in range(2):
            work = rk.clone()
            ctx.sequence(rv)
            if rep == 1:
                ctx.profile_reset()
            ctx.sort_pairs(work, rv)
        ctx.sync()
        radix_stats = (*ctx.profile_get("sort_pass"), ctx.profile_spread("sort_pass"), *ctx.profile_get("sort_pass_iota"),
                       ctx.profile_spread("sort_pass_iota"),
                       f"cstone_hip_sort_pairs of {n_sorted} random pairs (this rank's share), outside the timed region")
        # (the timed syncs' own pass launches include the small sorts: they stay out of the kernel table)
        sort_ms_in_syncs = (timed_stages["sort_pass"][0] + timed_stages["sort_pass_iota"][0]) / args.steps
        timed_stages["sort_pass"] = timed_stages["sort_pass_iota"] = (0.0, 0)
        del rk, rv, work
        invariants_ok = pipe.invariants(n_local * world)
    # what the output says about the headline pipeline (later sections replace `pipe`)
    head = {"f_leaves": pipe.f_leaves, "g_leaves": pipe.g_leaves,
            "n_scratch": len(pipe.scratch) if hasattr(pipe, "scratch") else 1}
    if distributed:
        head.update(assigned=pipe.assigned, halos=pipe.halos, stats=dict(pipe.stats), transport=pipe.transport,
                    rccl_ranks=pipe.rccl_ranks, resorts=int(pipe.dom.view().resorts))
    extras = {}
    if not distributed and not args.no_variants:
        # the same syncs with the radix sort forced over ALL key digits (what the reference's GPU path does every time;
        # by default Domain::sync sorts the digits above the previous tree's leaf level and finishes the rest in runs)
        def timed_variant(before_step=None):
            """args.steps syncs (the motion before each of them outside the timed intervals) with their stage times and
            what the domain's counters say about them"""
            ctx.profile_reset()
            st0 = pipe.dom.stats()
            dt = run_syncs(args.steps, before_step) / args.steps
            st1 = pipe.dom.stats()
            nonlocal radix_stats
            if radix_stats is None and ctx.profile_get("sort_pass")[1]:
                radix_stats = (*ctx.profile_get("sort_pass"), ctx.profile_spread("sort_pass"),
                               *ctx.profile_get("sort_pass_iota"), ctx.profile_spread("sort_pass_iota"),
                               "the digit passes of the syncs of extras.all_digits_sorted (this run, radix sort forced)")
            return {"ms_per_step": dt * 1e3, "value": n_local / dt, "unit": "particles/s",
                    "stage_ms_per_step": {k: round(ctx.profile_get(k)[0] / args.steps, 4) for k in cstone_amd.STAGES
                                          if ctx.profile_get(k)[1]},
                    "syncs": {k: st1[k] - st0[k] for k in st1 if k != "last_movers"} | {"last_movers": st1["last_movers"]}}

        # the headline loop on ONE stream: the gather of x, y, z not next to the tree update but behind it -- what the
        # gather takes when nothing shares the memory system with it (the library reads the switch at every sync)
        os.environ["CSTONE_NO_GATHER_OVERLAP"] = "1"
        run_syncs(1, move)
        extras["one_stream"] = timed_variant(move)
        extras["one_stream"]["note"] = ("the headline loop with CSTONE_NO_GATHER_OVERLAP=1: every kernel of a sync on the "
                                        "context's one stream")
        del os.environ["CSTONE_NO_GATHER_OVERLAP"]
        run_syncs(1, move)
        # the same time-stepping loop with the radix sort forced over ALL key digits (what the reference's GPU path does
        # every time) ...
        pipe.dom.set_sort_mode(pipe.dom.SORT_ALL_DIGITS)  # cstone_hip_domain_set_sort_mode
        run_syncs(1, move)
        extras["all_digits_sorted"] = timed_variant(move)
        extras["all_digits_sorted"]["note"] = "the headline loop, every sync sorted from scratch over all 8 key digits"
        # ... and sorted from scratch the cheaper way: radix passes over the digits above the previous tree's leaf level +
        # run fix-up, no use of the previous order
        pipe.dom.set_sort_mode(pipe.dom.SORT_FROM_SCRATCH)
        run_syncs(1, move)
        extras["sorted_from_scratch"] = timed_variant(move)
        extras["sorted_from_scratch"]["note"] = "the headline loop without the incremental re-sort"
        pipe.dom.set_sort_mode(pipe.dom.SORT_INCREMENTAL)
        run_syncs(1, move)
        # other motions: 1 % of the particles displaced by up to 2h, and none at all (every sync gets back exactly what
        # the previous one returned: the friendliest input)
        run_syncs(1, pipe.jiggle)
        extras["moving_particles"] = timed_variant(pipe.jiggle)
        extras["moving_particles"]["note"] = "1% of the particles displaced by <= 2h before every sync"
        run_syncs(2, None)
        extras["zero_motion"] = timed_variant(None)
        extras["zero_motion"]["note"] = "no particle moves between the syncs"
        # the case the speculative box does not like: open boundaries whose outermost particles move, so the global box
        # changes with every sync, every key with it, and nothing of the previous order can be used
        run_syncs(2, pipe.drift_unclamped)
        extras["open_box_moving_extremes"] = timed_variant(pipe.drift_unclamped)
        extras["open_box_moving_extremes"]["note"] = (
            "every particle displaced by <= 0.1 h and NOT kept inside [0, 1]: the box changes with every sync (syncs.box_redos "
            "= speculative encodes thrown away; after one of them the extents are measured first), all keys change, the "
            "radix path sorts")
    if not distributed and args.neighbor_targets > 0:
        extras["find_neighbors"] = [pipe.find_neighbors(args.neighbor_targets, 0),
                                    pipe.find_neighbors(args.neighbor_targets, 128)]
    if not distributed and not args.no_plummer:
        # BASELINE configs[1]: 10^7 uniform particles, encode + radix sort (all digits) + cornerstone tree from the root +
        # linked octree, no halos, every stage through its own C-ABI entry (the kernel-level seam of the reference)
        n1 = min(10_000_000, n_local)
        g1 = torch.Generator(device=ctx.device).manual_seed(3)
        rdt = torch.float64 if args.real_bits == 64 else torch.float32
        cv = cstone_amd.HILBERT if args.curve == "hilbert" else cstone_amd.MORTON
        xs = [torch.rand(n1, dtype=rdt, device=ctx.device, generator=g1) for _ in range(3)]
        order = torch.empty(n1, dtype=torch.int32, device=ctx.device)
        unit = cstone_amd.make_cbox([0, 1] * 3)

        def config1():
            k = ctx.compute_sfc_keys(cv, args.key_bits, xs[0], xs[1], xs[2], unit)
            ctx.sequence(order)
            ctx.sort_pairs(k, order)
            tree, counts, _ = ctx.compute_octree(k, args.bucket_focus)
            ctx.build_octree(tree, num_leaves=counts.numel())
            return counts.numel()

        config1()
        barrier()
        t5 = time.perf_counter()
        for _ in range(args.steps):
            leaves1 = config1()
        barrier()
        per1 = (time.perf_counter() - t5) / args.steps
        extras["encode_sort_tree_1e7"] = {"workload": f"{n1:.0e} uniform particles: compute_sfc_keys + sort_pairs over all "
                                                      f"digits + compute_octree from the root (bucket {args.bucket_focus}) + "
                                                      "build_octree, no halos (BASELINE configs[1])",
                                          "ms_per_step": per1 * 1e3, "value": n1 / per1, "unit": "particles/s",
                                          "leaves": leaves1}
        del xs, order
        # BASELINE configs[2]: the same number of Plummer-sphere particles (the reference's recipe, deep and very uneven
        # tree), full Domain::sync as a client runs it -- tight box measured by the domain, EVERY particle displaced by
        # <= 0.1 h before every sync and NOT clamped, so the outermost particles move the open box -- and findNeighbors on
        # the domain's octree; reported next to the headline number, not part of it
        del pipe
        torch.cuda.empty_cache()
        t2 = time.perf_counter()
        pl = SyncPipeline(ctx, n_local, args.key_bits, args.real_bits, args.curve, bucket_global, args.bucket_focus,
                          seed=7, dist="plummer")
        torch.cuda.synchronize()
        gen_s = time.perf_counter() - t2
        pipe = pl  # (run_syncs steps `pipe`)
        barrier()
        t3 = time.perf_counter()
        pl.first_sync()
        barrier()
        first_pl = time.perf_counter() - t3
        run_syncs(2, pl.drift)
        st0 = pl.dom.stats()
        ctx.profile_enable(0 if args.no_stage_timers else 1)
        ctx.profile_reset()
        per = run_syncs(args.steps, pl.drift) / args.steps
        st1 = pl.dom.stats()
        pl_stage = {k: round(ctx.profile_get(k)[0] / args.steps, 4) for k in cstone_amd.STAGES if ctx.profile_get(k)[1]}
        pl.note_leaves()
        vb = pl.dom.view().box
        lim = [float(v) for v in list(vb.lim)]
        run_syncs(2, None)
        st2 = pl.dom.stats()
        per0 = run_syncs(args.steps, None) / args.steps
        st3 = pl.dom.stats()
        counters = ("syncs", "resorts", "resort_fallbacks", "box_redos", "full_sort_fallbacks")
        extras["plummer"] = {"workload": f"{n_local:.0e} particles of the reference's Plummer sphere (test/coord_samples/"
                                         f"plummer.hpp restated: srand48(42), R < 100, scale 3 pi / 16, centre of mass at the "
                                         f"origin), h = half the radius that holds 100 particles at the local density, "
                                         f"bucketFocus {args.bucket_focus}, tight open box; before every sync EVERY particle is "
                                         f"displaced by <= 0.1 h per coordinate, not clamped",
                             "generation_s": gen_s, "first_sync_ms": first_pl * 1e3, "ms_per_step": per * 1e3,
                             "value": n_local / per, "unit": "particles/s", "focus_leaves": pl.f_leaves, "box": lim,
                             "syncs": {k: st1[k] - st0[k] for k in counters} | {"last_movers": st1["last_movers"]},
                             "stage_ms_per_step": pl_stage,
                             "zero_motion": {"ms_per_step": per0 * 1e3, "value": n_local / per0,
                                             "syncs": {k: st3[k] - st2[k] for k in counters},
                                             "note": "the same cloud with nothing moving between the syncs (round 3's number)"},
                             "find_neighbors": pl.find_neighbors(args.neighbor_targets, 0)
                             if args.neighbor_targets > 0 else None}
    if not distributed and world == 1 and not args.no_mr_extra:
        # the curve's first point must be comparable with the others: the same cloud through the code path the N-GPU runs
        # take (cstone_hip_domain_mr_sync, collectives served by RCCL with a communicator of ONE rank)
        keep = pipe
        pipe = None
        del keep
        torch.cuda.empty_cache()
        init_group()
        mr = DistributedPipeline(ctx, n_local, n_global, args.key_bits, args.real_bits, args.curve, bucket_global,
                                 args.bucket_focus, seed=42, dist=args.dist, rank=0, world=1, backend=backend)
        pipe = mr
        distributed = True  # (barrier() now includes the process group's)
        barrier()
        t6 = time.perf_counter()
        mr.first_sync()
        barrier()
        first_mr = time.perf_counter() - t6
        run_syncs(args.warmup, mr.drift)
        ctx.profile_enable(0)
        dt = run_syncs(args.steps, mr.drift) / args.steps
        ctx.profile_enable(0 if args.no_stage_timers else 1)
        ctx.profile_reset()
        run_syncs(args.steps, mr.drift)
        mr_stage = {k: round(ctx.profile_get(k)[0] / args.steps, 4) for k in cstone_amd.STAGES if ctx.profile_get(k)[1]}
        extras["mr_path_world_of_one"] = {
            "workload": "the headline cloud and motion through cstone_hip_domain_mr_sync (the N-GPU code path) on one rank",
            "ms_per_step": dt * 1e3, "value": n_local / dt, "unit": "particles/s", "first_sync_ms": first_mr * 1e3,
            "stage_ms_per_step": mr_stage, "transport": mr.transport, "rccl_ranks": mr.rccl_ranks,
            "syncs_resorted_total": int(mr.dom.view().resorts), "invariants_ok": mr.invariants(n_local)}
        distributed = False
        pipe = None
        del mr
        torch.distributed.destroy_process_group()
    ctx.profile_enable(False)
    ctx.sync()  # raises if a device-side check tripped

    if rank == 0:
        kbytes, rbytes = args.key_bits // 8, args.real_bits // 8
        n_scratch = head["n_scratch"]
        # ---- roofline.  Every kernel that moves the particle arrays, with its ALGORITHMIC bytes per launch (DESIGN.md
        # section 3) over its launch time measured live with HIP events on the context's stream (stage timers).  The
        # top-level fields are those of the kernel with the largest total time inside the timed syncs.
        models = [  # (kernel, stage, bytes per particle and launch, what the bytes are)
            ("encodeResortKernel" if stage_ms.get("resort_leaves", 0) > 0 else "encodeHistogramKernel", "encode",
             3 * rbytes + 2 * kbytes, "x, y, z and the old key read, the new key written"),
            (("leafSortWaveKernel<LeafFields> (keys + x, y, z, h in one pass)", "resort_leaves", 2 * kbytes + 4 + 8 * rbytes,
              "key, x, y, z, h read; key, old index, x, y, z, h written at the leaf's new place (four scratch arrays: the "
              "field-carrying leaf pass, no gather passes; opt-in, DESIGN.md section 10)")
             if n_scratch >= 4 and os.environ.get("CSTONE_FUSED_LEAF_PASS") is not None else
             ("leafSortWaveKernel", "resort_leaves", 2 * kbytes + 4,
              "key read; key + old index written (one wave per leaf; with fewer movers than tiles a launch of "
              "leafSortKernel for the quiet tiles comes first, inside the same bracket)")),
            ("onesweepKernel", "sort_pass", 2 * (kbytes + 4), "key + index read and written"),
            ("onesweepKernel (positions generated)", "sort_pass_iota", 2 * kbytes + 4, "key read; key + index written"),
            (("gatherMultiKernel (x, y, z in one launch)", "gather", 4 + 6 * rbytes,
              "index read once, three elements read and written") if n_scratch >= 3 else
             ("gatherKernel", "gather", 4 + 2 * rbytes, "index + element read, element written; one launch per array")),
            ("gatherHaloRadiiKernel", "gather_h", 4 + 2 * rbytes, "index + h read, h written (+ one radius per leaf)"),
            ("placeColumnsKernel", "place", 4 + 2 * kbytes + 8 * rbytes,
             "multi-rank sync: index read; key, x, y, z, h read and written to their final slots in one pass"),
        ]
        if distributed:
            # a rank's sync also runs the element-wise gather / scatter and the pass kernel on small inputs (newcomers, tree
            # node keys): their stage averages say nothing about the full-size launches and stay out of the table
            models = [m for m in models if m[1] not in ("gather", "sort_pass", "sort_pass_iota")]
        tjson = None
        for tname in ("r04_kernel_hbm_traffic.json", "r03_kernel_hbm_traffic.json", "r02_kernel_hbm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and tjson is None:
                tjson = (tname, json.load(open(tpath)))

        def pmc_traffic(kernel):
            """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled + WRITE_SIZE), scaled to
            this run's particles per launch; None when the profile does not hold the kernel"""
            if not tjson:
                return None
            wanted = kernel.split(" ")[0]
            for row in tjson[1]["kernels"]:
                if row["kernel"] == wanted:
                    return (row["hbm_read_bytes"] + row["hbm_write_bytes"]) * n_sorted / tjson[1].get("particles", 1e8)
            return None

        table = []
        for kernel, stage, bpp, what in models:
            ms, launches = timed_stages.get(stage, (0.0, 0))
            if not launches:
                continue
            lo, med, hi = timed_spreads[stage]
            bytes_launch = float(bpp) * n_sorted
            gbs = bytes_launch * launches / (ms * 1e-3) / 1e9
            table.append({"kernel": kernel, "stage": stage, "launches": launches, "bytes_per_particle": bpp,
                          "bytes": what, "bytes_per_launch": bytes_launch, "avg_ms": ms / launches, "min_ms": lo,
                          "median_ms": med, "max_ms": hi, "total_ms_per_step": ms / args.steps, "achieved": gbs,
                          "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(kernel)})
            alone = extras.get("one_stream", {}).get("stage_ms_per_step", {}).get(stage)
            if stage == "gather" and n_scratch >= 3 and alone and not distributed:
                # in the timed syncs this launch runs on the context's second stream next to the tree update, which takes
                # bandwidth from it; on its own (extras.one_stream, the same run):
                table[-1]["concurrent"] = "on the second stream, next to the tree update and the linked-octree build"
                table[-1]["alone"] = {"avg_ms": alone, "achieved": bytes_launch / (alone * 1e-3) / 1e9,
                                      "frac": bytes_launch / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "measured_on": "extras.one_stream"}
        top = max(table, key=lambda r: r["total_ms_per_step"]) if table else None
        sync_bytes = sum(r["bytes_per_launch"] * r["launches"] for r in table) / args.steps
        # the radix pass the north star singles out: not part of a steady-state sync any more (the re-sort replaces it);
        # measured live in this run on the syncs of extras.sorted_from_scratch / all_digits_sorted
        onesweep = None
        if radix_stats:
            pass_ms, pass_launches, pass_spread, iota_ms, iota_launches, iota_spread, radix_source = radix_stats
            per_launch_bytes = 2.0 * (kbytes + 4) * n_sorted
            iota_launch_bytes = (2.0 * kbytes + 4) * n_sorted
            total_bytes = per_launch_bytes * pass_launches + iota_launch_bytes * iota_launches
            total_s = (pass_ms + iota_ms) * 1e-3
            ach = total_bytes / total_s / 1e9 if total_s > 0 else 0.0
            onesweep = {"kernel": "onesweepKernel (one 8-bit radix pass over key+index pairs)", "achieved": ach,
                        "frac": ach / HBM_PEAK_GBS, "launches": pass_launches + iota_launches,
                        "avg_launch_ms": total_s * 1e3 / max(1, pass_launches + iota_launches),
                        "traffic": pmc_traffic("onesweepKernel"),
                        "per_pass": {"regular": {"launches": pass_launches, "bytes_per_launch": per_launch_bytes,
                                                 "avg_ms": pass_ms / max(1, pass_launches), "min_ms": pass_spread[0],
                                                 "median_ms": pass_spread[1], "max_ms": pass_spread[2]},
                                     "positions_generated": {"launches": iota_launches,
                                                             "bytes_per_launch": iota_launch_bytes,
                                                             "avg_ms": iota_ms / max(1, iota_launches),
                                                             "min_ms": iota_spread[0], "median_ms": iota_spread[1],
                                                             "max_ms": iota_spread[2]}},
                        "measured_on": radix_source}
            ad = extras.get("all_digits_sorted")
            if ad and not distributed:
                # SURVEY section 8(d)'s model of the WHOLE sort of (key, index) pairs: the keys read once for the digit
                # histograms + P digit passes of 2 (K + 4) bytes = K + P * 2 * (K + 4) bytes per pair (200 at K = 8, P = 8).
                # Measured on extras.all_digits_sorted: the digit passes, the pass that generates the positions and the
                # histogram step; the digits are COUNTED inside the encode kernel there (its extra time over the encode of
                # the headline's syncs is added)
                st = ad["stage_ms_per_step"]
                counting = max(0.0, st.get("encode", 0.0) - stage_ms.get("encode", 0.0))
                sort_ms = st.get("sort_pass", 0.0) + st.get("sort_pass_iota", 0.0) + st.get("sort_hist", 0.0) + counting
                passes = kbytes
                bytes_pair = kbytes + passes * 2 * (kbytes + 4)
                if sort_ms > 0:
                    onesweep["whole_sort"] = {"model_bytes_per_pair": bytes_pair, "passes": passes, "ms": sort_ms,
                                              "digit_counting_inside_the_encode_ms": counting,
                                              "achieved": bytes_pair * n_sorted / (sort_ms * 1e-3) / 1e9,
                                              "frac": bytes_pair * n_sorted / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "measured_on": "extras.all_digits_sorted (all digit passes of one sync)"}
        if distributed and onesweep and sort_ms_in_syncs > (top["total_ms_per_step"] if top else 0.0):
            # several ranks below the re-sort's size: the digit passes are the largest share of a rank's sync.  Their stage
            # also holds the small sorts, so the kernel is priced on this rank's share of the particles measured on its own
            top = {"kernel": onesweep["kernel"], "achieved": onesweep["achieved"], "frac": onesweep["frac"],
                   "traffic": onesweep["traffic"], "bytes_per_launch": onesweep["per_pass"]["regular"]["bytes_per_launch"],
                   "avg_ms": onesweep["avg_launch_ms"], "launches": onesweep["launches"]}
        roofline = {"bound": "hbm", "kernel": top["kernel"] if top else None,
                    "achieved": top["achieved"] if top else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": top["frac"] if top else 0.0, "traffic": top["traffic"] if top else None,
                    "traffic_source": (f"profiles/{tjson[0]} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                       f"bench.py, not this run)") if tjson else None,
                    "bytes_per_launch": top["bytes_per_launch"] if top else None,
                    "avg_launch_ms": top["avg_ms"] if top else None, "launches": top["launches"] if top else 0,
                    "dominant_by": "total time of the kernel's launches inside the timed syncs",
                    "measured_on": "the timed syncs (HIP events around every launch, on the context's stream)",
                    "kernels": table,
                    "sync": {"algorithmic_bytes_per_step": sync_bytes,
                             "achieved": sync_bytes / (elapsed / args.steps) / 1e9,
                             "frac": sync_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                             "note": "bytes of the kernels listed here (x, y, z, h gathers included) over the wall time of a "
                                     "whole sync"},
                    "onesweep": onesweep}
        out = {
            "metric": "particles/sec domain.sync (encode+sort+tree+halo), 10^8 uniform, 1/2/4/8 GPU",
            "value": n_local * world * args.steps / elapsed,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": f"u{args.key_bits}/f{args.real_bits}",
            "data": "synthetic",
            "config": {"workload": f"{n_global:.0e} {args.dist} particles, {args.key_bits}-bit {args.curve} keys, "
                                   f"f{args.real_bits} coordinates, bucketFocus {args.bucket_focus}, "
                                   f"bucket {bucket_global}, time-stepping loop: before every sync EVERY particle is "
                                   f"displaced by <= 0.1 h per coordinate (outside the timed intervals, each sync is "
                                   f"bracketed on its own); the arrays a sync returns go into the next one; particles "
                                   f"still inside their leaf are ordered leaf by leaf, the others are binned "
                                   f"(csrc/resort.hpp) -- the same order as sorting all keys; extras: the radix sort "
                                   f"instead (sorted_from_scratch / all_digits_sorted), other motions (moving_particles, "
                                   f"zero_motion)"
                                   + ("" if not distributed else
                                      f"; {world} rank(s): SFC domain decomposition, particle exchange, locally essential "
                                      f"tree and halo exchange with all_to_all over RCCL"),
                       "particles_per_gpu": n_local, "focus_leaves": head["f_leaves"], "global_leaves": head["g_leaves"],
                       "path": path, "dist": args.dist, "self_launched": os.environ.get("CSTONE_BENCH_SELF_LAUNCHED") == "1",
                       **({"syncs_timed": timed_stats} if timed_stats else {}),
                       **({"invariants_ok": invariants_ok, "rank0_assigned": head["assigned"], "rank0_halos": head["halos"],
                           "rank0_exchange": head["stats"], "rccl_ranks": head["rccl_ranks"],
                           "rank0_syncs_resorted_total": head["resorts"],
                           "orchestration": "libcstone_hip (cstone_hip_domain_mr_sync)",
                           "transport": head["transport"]} if distributed else {})},
            "roofline": roofline,
            "stage_ms_per_step": stage_ms,
            "first_sync_ms": first_sync_ms,
            "extras": extras,
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(int(args.cpu_sample), args.key_bits, args.real_bits, args.curve,
                                               args.bucket_focus)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

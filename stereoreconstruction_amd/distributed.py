"""Multi-GPU plumbing of the path: stereo pairs / views are independent units, sharded over
one process per GPU; the only exchange is the gather of per-view depth maps (RCCL over xGMI
when the tensors live on the GPU, gloo on CPU in the tests).  SURVEY.md section 8(e)."""
import torch
import torch.distributed as dist


def shard_units(n_units, world_size, rank):
    """Contiguous, balanced split of units 0..n_units-1; returns range(lo, hi) for `rank`.
    The first (n_units % world_size) ranks take one extra unit."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def gather_depth_maps(local, dst=0):
    """Gather equally-shaped depth-map tensors to rank `dst`; returns the list (rank order) there,
    None elsewhere.  With world_size 1 returns [local]."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def all_gather_depth_maps(local):
    """Every rank receives every rank's maps (MVS cross-check needs all views' maps)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local)
    return out


class HipMultiViewEngine:
    """The per-rank side of multiview_sharded() on a GPU: one srh_context holding all views
    (images and cameras are replicated -- a few MB -- so that every rank can match its own views
    against any neighbour), depth maps exchanged as device tensors.  Views may differ in raster size
    (MultiViewStereo keeps one VectorImage per view, multiviewstereo.cpp:216-244): maps travel
    flattened and padded with NaN to the largest view, `shape` is that common (1-D) exchange shape."""

    def __init__(self, ctx, slots, neighbours, params, device):
        self.ctx, self.slots, self.neigh, self.p, self.device = ctx, list(slots), neighbours, params, device
        self.sizes = [ctx.view_size(s) for s in self.slots]     # (w, h) per view
        self.shape = (max(w * h for (w, h) in self.sizes),)

    def initial_estimate(self, v):
        self.ctx.mvs_initial_estimate(self.slots[v], [self.slots[n] for n in self.neigh[v]], self.p)

    def depth_tensor(self, v):
        w, h = self.sizes[v]
        if torch.device(self.device).type == "cpu":             # host exchange (gloo rehearsal of the N>1 path)
            t = torch.full(self.shape, float("nan"), dtype=torch.float64)
            t[:w * h] = torch.from_numpy(self.ctx.download_depth(self.slots[v])).reshape(-1)
            return t
        t = torch.full(self.shape, float("nan"), dtype=torch.float64, device=self.device)
        self.fence()                                             # the fill runs on torch's stream, the copy on the library's
        self.ctx.copy_depth_to_device(self.slots[v], t.data_ptr(), t.numel() * 8)
        return t

    def set_depth(self, v, t):
        assert t.is_contiguous() and t.dtype == torch.float64 and tuple(t.shape) == self.shape
        w, h = self.sizes[v]
        if t.device.type == "cpu":
            self.ctx.upload_depth(self.slots[v], t[:w * h].reshape(h, w).numpy())
        else:
            self.ctx.copy_depth_from_device(self.slots[v], t.data_ptr(), t.numel() * 8)

    def cross_check(self, v):
        self.ctx.mvs_cross_check(self.slots, v, self.p)

    def fence(self):
        # the library works on its own stream: order it against torch's streams / RCCL, both ways
        self.ctx.synchronize()
        if torch.device(self.device).type != "cpu":
            torch.cuda.synchronize(self.device)


def multiview_sharded(engine, n_views):
    """MultiViewStereo::runTask (multiviewstereo.cpp:323-447) over the ranks of the default process
    group.  Views are sharded (shard_units); each rank runs computeInitialEstimate for its own views;
    ONE all-gather moves every rank's maps to every rank (the cross-check of a view reads all other
    views' maps, multiviewstereo.cpp:694-719); then every rank runs the same sequential, order-dependent
    cross-check chain, so the final maps are resident everywhere, rank 0 included.  Returns the list of
    views this rank estimated.  `engine`: initial_estimate(v), depth_tensor(v), set_depth(v, tensor),
    cross_check(v), fence()."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = list(shard_units(n_views, world, rank))
    for v in mine:
        engine.initial_estimate(v)
    if world > 1:
        per = -(-n_views // world)                               # equal-sized contributions: pad the short ranks
        maps = [engine.depth_tensor(v) for v in mine]
        maps += [torch.full(engine.shape, float("nan"), dtype=torch.float64, device=engine.device)
                 for _ in range(per - len(maps))]
        engine.fence()
        local = torch.stack(maps)
        everyone = all_gather_depth_maps(local)
        # RCCL only makes torch's current stream wait for the collective; the copies below run on the library's own
        # stream: order the two before the first map is read (`everyone` stays alive until the final fence)
        engine.fence()
        for r in range(world):
            if r == rank:
                continue
            for k, v in enumerate(shard_units(n_views, world, r)):
                engine.set_depth(v, everyone[r][k].contiguous())
        engine.fence()
    for v in range(n_views):                                     # in view order: each view reads the earlier, filtered maps
        engine.cross_check(v)
    engine.fence()
    return mine


def twoview_rowbands_sharded(engine, height):
    """ONE TwoViewStereo pair over the ranks of the default process group by row bands (BASELINE.md's
    C3 row "+ row-band split for 2/4/8"): computeCostVolumes is independent per reference row
    (twoviewstereo.cpp:265-332, 436-500 -- the row loops the reference itself hands to OpenMP), so rank r computes
    rows shard_units(height, world, r) of BOTH maps; the bands are gathered on rank 0 (padded to the tallest band),
    stitched, and the order-dependent cross-check (twoviewstereo.cpp:596-672: left pass, then right pass on the
    filtered left map) runs there on the whole maps.  Returns (y0, y1) of this rank's band.
    `engine`: wta_rows(y0, y1) -- both directions for rows [y0, y1); band_tensor(view, y0, y1, rows) -> (rows, W)
    float64 tensor padded with NaN; set_band(view, y0, y1, tensor); cross_check(); fence(); device."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_units(height, world, rank)
    y0, y1 = mine.start, mine.stop
    if y1 > y0:
        engine.wta_rows(y0, y1)
    if world > 1:
        tallest = -(-height // world)
        local = torch.stack([engine.band_tensor(view, y0, y1, tallest) for view in (0, 1)])
        engine.fence()
        bands = gather_depth_maps(local, dst=0)
        engine.fence()
        if rank == 0:
            for r in range(1, world):
                rr = shard_units(height, world, r)
                if rr.stop > rr.start:
                    for view in (0, 1):
                        engine.set_band(view, rr.start, rr.stop, bands[r][view])
            engine.fence()
    if rank == 0:
        engine.cross_check()
    engine.fence()
    return y0, y1


class HipTwoViewBandEngine:
    """The per-rank side of twoview_rowbands_sharded() on a GPU: one srh_context holding both views (slots 0, 1).
    Bands travel as device tensors (RCCL; the maps never leave the device: srh_view_depth_copy_to/from_device) or,
    device "cpu", through host memory (the gloo rehearsal)."""

    def __init__(self, ctx, params, device):
        self.ctx, self.p, self.device = ctx, params, device
        self.w, self.h = ctx.view_size(0)
        self.on_dev = torch.device(device).type != "cpu"
        self.full = [None, None]            # this rank's maps as tensors, bands of other ranks written into them
        self.dirty = [False, False]

    def wta_rows(self, y0, y1):
        self.ctx.twoview_wta(0, 1, self.p, y0, y1)
        self.ctx.twoview_wta(1, 0, self.p, y0, y1)

    def _map(self, view):
        if self.full[view] is None:
            if self.on_dev:
                t = torch.empty((self.h, self.w), dtype=torch.float64, device=self.device)
                self.fence()                                     # the allocation is torch's, the copy the library's stream
                self.ctx.copy_depth_to_device(view, t.data_ptr(), t.numel() * 8)
                self.ctx.synchronize()
            else:
                t = torch.from_numpy(self.ctx.download_depth(view))
            self.full[view] = t
        return self.full[view]

    def band_tensor(self, view, y0, y1, rows):
        self.full[view] = None                                   # a fresh pass: re-read the map
        t = torch.full((rows, self.w), float("nan"), dtype=torch.float64, device=self.device if self.on_dev else "cpu")
        if y1 > y0:
            t[:y1 - y0] = self._map(view)[y0:y1]
        return t

    def set_band(self, view, y0, y1, t):
        self._map(view)[y0:y1] = t[:y1 - y0]
        self.dirty[view] = True

    def cross_check(self):
        for view in (0, 1):
            if self.dirty[view]:
                m = self.full[view]
                if self.on_dev:
                    torch.cuda.synchronize(self.device)
                    self.ctx.copy_depth_from_device(view, m.data_ptr(), m.numel() * 8)
                else:
                    self.ctx.upload_depth(view, m.numpy())
                self.dirty[view] = False
        self.ctx.twoview_cross_check(0, 1, self.p)

    def fence(self):
        self.ctx.synchronize()
        if self.on_dev:
            torch.cuda.synchronize(self.device)

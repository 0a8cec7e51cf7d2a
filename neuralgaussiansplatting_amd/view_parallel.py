"""View-sharded data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:76-78,
utils/general_utils.py:133); this module adds the multi-GPU form: Gaussian
parameters are replicated, each rank (one process per GPU) rasterizes its own
camera view forward and backward, and the per-Gaussian gradients of all ranks
are summed by ONE all-reduce per step (RCCL over xGMI with the ``nccl``
backend; ``gloo`` works on CPU tensors for tests).

Two forms of the exchange are provided; both leave every rank with the sum over
all views of the five parameter gradients.

``GradientBucket`` (dense): the backward pass writes its five parameter
gradients straight into slices of one flat bucket, which is all-reduced:

    [ dL_dmeans3D 3P | dL_dsh 3MP | dL_dopacity P | dL_dscales 3P | dL_drotations 4P ]

``FactoredGradientExchange``: 3M of those 3M+11 floats per Gaussian are the SH
coefficient gradient, which for one view is just basis(view direction) x (the
clamp-masked colour gradient, 3 floats).  So each rank contributes its masked
colour gradient and camera position to an all-gather (3 floats per Gaussian and
view instead of 3M), the other 11 floats go through a small all-reduce, and every
rank rebuilds the summed SH gradient locally, adding the views in rank order.
On 8 GPUs at M = 16 that moves about 2.6x fewer bytes per GPU than the dense
all-reduce, and all ranks hold bit-identical SH gradients.

Both forms take ``chunks``: the per-Gaussian stage of the backward pass is then
queued in that many runs of Gaussians, the buffers are laid out run by run, and
each run's slice leaves in a collective of its own as soon as its kernel has
been queued -- the exchange travels while the rest of the gradients are still
being computed (default: 4 runs whenever there is more than one rank).

Both forms take ``views_per_rank``: a rank then renders that many views per
step, every view's collectives are queued as soon as its backward pass is
(``post_view``) and travel while the next view is rendered, and ``finish``
waits once and adds the views up.  With ``stats`` the two sums of the
densification statistics (scene/gaussian_model.py:550-552) ride in the same
buffers, so after a step every rank holds their totals over all views.
"""
import torch
import torch.distributed as dist

from .diff_gaussian_rasterization import _lib as _C

BUCKET_FIELDS = ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations")
# densification statistics in summable form (see ``_C.view_statistics``): they ride in the gradients' all-reduce
STAT_FIELDS = ("stat_norm", "stat_vis")


def _all_reduce_sum(t, group=None):
    """Sum ``t`` over the ranks in place.  RCCL works on the device tensor; with gloo (CPU tests, or several
    ranks rehearsing on one GPU) a device tensor is staged through host memory."""
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather_rows(rows, local, group=None):
    """rows[r] <- rank r's ``local`` (rows is [world, n], contiguous)."""
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(rows.view(-1), local, group=group)
    elif rows.is_cuda:
        h = torch.empty(rows.shape, dtype=rows.dtype)
        dist.all_gather(list(h.unbind(0)), local.cpu(), group=group)
        rows.copy_(h)
    else:
        dist.all_gather(list(rows.unbind(0)), local, group=group)


def _aligned_fields(names, sizes):
    """Fields laid end to end, each starting on a 16-byte boundary (the kernels that read and write them use
    four-float accesses); the few padding floats stay zero.  -> ({field: (offset, length)}, total) in floats."""
    lay, off = {}, 0
    for name, n in zip(names, sizes):
        lay[name] = (off, n)
        off += (n + 3) & ~3
    return lay, off


def chunk_bounds(P, chunks):
    """[(first, count)] runs of Gaussians that tile [0, P); every run but the last is a multiple of 256 long (the
    per-Gaussian kernels work in blocks of 256)."""
    chunks = max(int(chunks or 1), 1)
    if chunks == 1 or P <= 256:
        return [(0, P)]
    per = ((P + chunks - 1) // chunks + 255) // 256 * 256
    return [(f, min(per, P - f)) for f in range(0, P, per)]


def chunked_layout(names, per_row, bounds):
    """Fields laid out run by run: all fields of the first run of Gaussians, then all fields of the second, ...
    -> ([{field: (offset, length)} per run], [(offset, length) of each run], total), in float32 elements."""
    per_chunk, spans, off = [], [], 0
    for first, count in bounds:
        lay, n = _aligned_fields(names, tuple(r * count for r in per_row))
        per_chunk.append({k: (off + o, ln) for k, (o, ln) in lay.items()})
        spans.append((off, n))
        off += n
    return per_chunk, spans, off


def bucket_layout(P, M, stats=False):
    """-> ({field: (offset, length)}, total) in float32 elements."""
    names, sizes = BUCKET_FIELDS, (3 * P, 3 * M * P, P, 3 * P, 4 * P)
    if stats:
        names, sizes = names + STAT_FIELDS, sizes + (P, P)
    return _aligned_fields(names, sizes)


def _grouped(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


_FIELD_ROW = dict(dL_dmeans3D=3, dL_dsh=None, dL_dopacity=1, dL_dscales=3, dL_drotations=4, stat_norm=1, stat_vis=1)
_SHAPES = dict(dL_dmeans3D=(-1, 3), dL_dopacity=(-1, 1), dL_dscales=(-1, 3), dL_drotations=(-1, 4), stat_norm=(-1, 1),
               stat_vis=(-1, 1))


class _ChunkedBuffers:
    """What the two exchange forms share: the run-by-run layout of their buffers and the bookkeeping of the
    collectives that leave from inside the backward pass."""

    def _init_layout(self, names, P, M, chunks, world):
        per_row = tuple(3 * M if n == "dL_dsh" else _FIELD_ROW[n] for n in names)
        if chunks is None:
            chunks = 4 if world > 1 else 1
        self.bounds = chunk_bounds(P, chunks)
        self.K = len(self.bounds)
        self.names = names
        self._lay, self._spans, total = chunked_layout(names, per_row, self.bounds)
        return total

    # Optional clock (benchmarks): per step three timing events on the compute stream -- when the first collective of
    # the step was handed its data, when the rank's own work of the step was all queued, when the last collective had
    # been waited for.  first -> done is how long the exchange took end to end, own -> done the part of it that
    # nothing of this rank's work covered.
    _clock = None

    def enable_clock(self, on=True):
        self._clock = [] if on else None
        self._clock_step = {}

    def _mark(self, key, once=False):
        if self._clock is None or (once and key in self._clock_step):
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._clock_step[key] = ev

    def _clock_close(self):
        if self._clock is not None:
            if "first" not in self._clock_step and "own" in self._clock_step:
                self._clock_step["first"] = self._clock_step["own"]
            self._clock.append(self._clock_step)
            self._clock_step = {}

    def clock_read(self):
        """-> (exchange ms end to end, exchange ms not covered by own work), averages over the steps since
        ``enable_clock``; synchronises the device."""
        if not self._clock:
            return None, None
        torch.cuda.synchronize()
        steps = [c for c in self._clock if {"first", "own", "done"} <= set(c)]
        if not steps:
            return None, None
        total = sum(c["first"].elapsed_time(c["done"]) for c in steps) / len(steps)
        exposed = sum(c["own"].elapsed_time(c["done"]) for c in steps) / len(steps)
        return total, exposed

    def _views_of(self, flat):
        """Per run of Gaussians: {field: 1-D view into ``flat``}."""
        return [{k: flat[o:o + n] for k, (o, n) in lay.items()} for lay in self._lay]

    def _shaped_from(self, chunk_views):
        out = {}
        for name in self.names:
            parts = [cv[name] for cv in chunk_views]
            t = parts[0] if len(parts) == 1 else torch.cat(parts)
            out[name] = t.view((-1, self.M, 3) if name == "dL_dsh" else _SHAPES[name])
        return out

    def chunk_views(self):
        """[(first, count, {field: tensor shaped [count, ...]})]: the step's sums run by run, views into the buffer
        (what an optimiser steps from; ``shaped()`` concatenates them, which copies when there are several runs)."""
        out = []
        for (first, count), cv in zip(self.bounds, self._result_chunks):
            d = {k: v.view((-1, self.M, 3) if k == "dL_dsh" else _SHAPES[k]) for k, v in cv.items()}
            out.append((first, count, d))
        return out

    def _plan(self, slot_chunks, skip, dL_dmeans2D, radii, max_radii2D, ready):
        """The ``chunks`` argument of ``_C.rasterize_gaussians_backward_ex``."""
        clist = []
        for (first, count), cv in zip(self.bounds, slot_chunks):
            c = dict(first=first, count=count)
            c.update({k: v for k, v in cv.items() if k not in STAT_FIELDS and k not in skip})
            clist.append(c)

        def on_chunk(k):
            first, count = self.bounds[k]
            if self.stats:  # the chunk's kernel has just been queued: its view-space gradients are on their way
                cv = slot_chunks[k]
                _C.view_statistics(dL_dmeans2D[first:first + count], radii[first:first + count], cv["stat_norm"],
                                   cv["stat_vis"], None if max_radii2D is None else max_radii2D[first:first + count])
            ready(k)

        return clist, on_chunk


class GradientBucket(_ChunkedBuffers):
    """One flat float32 buffer holding every parameter gradient of a step, summed over the ranks -- by ONE
    all-reduce, or with ``chunks`` > 1 by one all-reduce per run of Gaussians, each leaving as soon as the backward
    pass has queued the kernel that fills it.

    With ``stats`` the two densification-statistics rows travel in the same buffer.  With ``views_per_rank`` > 1
    each view's backward pass writes into a second buffer of the same layout and ``post_view`` adds it to the
    bucket, so the step still ends with one all-reduce of one buffer."""
    skip_sh = False  # the backward pass produces dL_dsh itself

    def __init__(self, P, M, device, stats=False, views_per_rank=1, chunks=None, group=None):
        self.P, self.M, self.stats, self.V, self.group = P, M, stats, int(views_per_rank), group
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        names = BUCKET_FIELDS + (STAT_FIELDS if stats else ())
        total = self._init_layout(names, P, M, chunks if self.V == 1 else 1, world)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self._result_chunks = self._views_of(self.flat)
        self.views = self._result_chunks[0] if self.K == 1 else None
        if self.V > 1:
            self.scratch = torch.zeros(total, dtype=torch.float32, device=device)
            self._scratch_chunks = self._views_of(self.scratch)
        self._on_gpu = torch.device(device).type == "cuda"
        self._side = torch.cuda.Stream(device=device) if self._on_gpu and self.K > 1 else None
        self._events = [torch.cuda.Event() for _ in range(self.K)] if self._side is not None else None
        self._pending, self._sent = [], set()

    chunked = property(lambda self: self.K > 1)

    def slot_views(self, slot=0):
        """Where the backward pass of this rank's view ``slot`` writes its gradients (one run of Gaussians only)."""
        return (self._scratch_chunks if self.V > 1 else self._result_chunks)[0]

    def slot_chunks(self, slot=0):
        """Per run of Gaussians: where the backward pass of this rank's view ``slot`` writes that run's gradients."""
        return self._scratch_chunks if self.V > 1 else self._result_chunks

    def begin_chunks(self, slot, dL_dmeans2D, radii, max_radii2D=None):
        return self._plan(self._result_chunks, (), dL_dmeans2D, radii, max_radii2D, self._chunk_ready)

    def _chunk_ready(self, k):
        """Chunk k of the bucket is final once what is queued on the current stream has run: send it."""
        if not (_grouped(self.group) and dist.get_backend(self.group) == "nccl"):
            return
        o, n = self._spans[k]
        self._mark("first", once=True)
        self._events[k].record(torch.cuda.current_stream(self.flat.device))
        self._side.wait_event(self._events[k])
        with torch.cuda.stream(self._side):
            self._pending.append(dist.all_reduce(self.flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
        self._sent.add(k)

    def shaped(self):
        return self._shaped_from(self._result_chunks)

    def write_statistics(self, slot, dL_dmeans2D, radii, max_radii2D=None):
        v = self.slot_views(slot)
        _C.view_statistics(dL_dmeans2D, radii, v["stat_norm"], v["stat_vis"], max_radii2D)

    def post_view(self, slot, scene=None, cam=None, aux=None, group=None):
        """After the backward pass of view ``slot``: fold it into the step's sum."""
        if self.V > 1:
            if slot == 0:
                self.flat.copy_(self.scratch)
            else:
                self.flat.add_(self.scratch)

    def finish(self, scene=None, group=None, sh_adam=None):
        if sh_adam is not None:
            raise RuntimeError("the dense exchange has no fused SH optimiser step")
        return self.all_reduce(group=group)

    def all_reduce(self, average=False, group=None, async_op=False):
        """Sum (or mean) the bucket over all ranks: whatever has not left from inside the backward pass leaves now,
        and everything in flight is waited for."""
        group = group if group is not None else self.group
        if not _grouped(group):
            return None
        if async_op and not self._sent:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        self._mark("own")
        if not self._sent:
            _all_reduce_sum(self.flat, group)
        else:
            for k, (o, n) in enumerate(self._spans):
                if k not in self._sent:
                    _all_reduce_sum(self.flat[o:o + n], group)
            for work in self._pending:
                work.wait()
        self._mark("done")
        self._clock_close()
        self._pending, self._sent = [], set()
        if average:
            self.flat.div_(dist.get_world_size(group))
        return None

    def bytes_sent_per_rank(self):
        return 4 * self.flat.numel()

    def reduce(self, scene=None, cam=None, aux=None, group=None, sh_adam=None):
        """One-view steps: ``post_view(0)`` and ``finish`` in one call (the dense form needs none of the arguments)."""
        self.post_view(0)
        return self.finish(scene, group, sh_adam)


REDUCED_FIELDS = ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")


class FactoredGradientExchange(_ChunkedBuffers):
    """Gradient exchange with the SH part factored into (masked colour gradient, camera position) per view.

    A rank renders ``views_per_rank`` views per step.  View k's backward pass writes into ``slot_views(k)``
    (``skip_sh`` tells it not to produce dL_dsh); ``post_view(k, ...)`` queues that view's collectives without
    waiting for them, so they travel while view k+1 is rendered; ``finish(scene)`` waits, adds the views up and
    rebuilds the SH gradient from all world x views_per_rank views.  Afterwards ``shaped()`` holds the sums over
    every view of every rank.  ``reduce(scene, cam, aux)`` = ``post_view(0, ...)`` + ``finish(...)`` for one view
    per rank.  With ``chunks`` > 1 the all-reduce of a view is split by runs of Gaussians and each part leaves from
    inside the backward pass (``begin_chunks``)."""
    skip_sh = True

    def __init__(self, P, M, device, world_size=None, group=None, stats=False, views_per_rank=1, chunks=None):
        self.P, self.M, self.group, self.stats, self.V = P, M, group, stats, int(views_per_rank)
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.world = world_size
        names = REDUCED_FIELDS + (STAT_FIELDS if stats else ())
        total = self._init_layout(names, P, M, chunks, world_size)
        V = self.V
        # one buffer per view in flight; `reduced` is where the step's total ends up (view 0's own buffer for V == 1)
        self.per_view = torch.zeros((V, total), dtype=torch.float32, device=device)
        self.reduced = self.per_view[0] if V == 1 else torch.zeros(total, dtype=torch.float32, device=device)
        self._result_chunks = self._views_of(self.reduced)
        self.views = self._result_chunks[0] if self.K == 1 else None
        self._slot_chunks = [self._views_of(self.per_view[k]) for k in range(V)]
        # one view's contribution: masked colour gradient [P,3], camera position, padding to whole 16 bytes
        self.row = ((3 * P + 3) + 3) & ~3
        self.local = torch.zeros((V, self.row), dtype=torch.float32, device=device)
        self.gathered = torch.zeros((V, self.world, self.row), dtype=torch.float32, device=device)
        self.dL_dsh = torch.empty((P, M, 3), dtype=torch.float32, device=device)
        # The masked colour gradient is final once the compositing backward has run; the backward pass hands it
        # over at that point (``early_color``) and the all-gather starts while the per-Gaussian stage still runs.
        self._on_gpu = torch.device(device).type == "cuda"
        self._color_event = [torch.cuda.Event() for _ in range(V)] if self._on_gpu else None
        self._chunk_event = [[torch.cuda.Event() for _ in range(self.K)] for _ in range(V)] if self._on_gpu else None
        self._side = torch.cuda.Stream(device=device) if self._on_gpu else None
        self._early = [False] * V
        self._pending = []
        self._sent = [set() for _ in range(V)]
        self._slot = 0

    chunked = property(lambda self: self.K > 1)

    def slot_views(self, slot=0):
        return self._slot_chunks[slot][0]

    def slot_chunks(self, slot=0):
        """Per run of Gaussians: where the backward pass of this rank's view ``slot`` writes that run's gradients."""
        return self._slot_chunks[slot]

    def begin_view(self, cam, slot=0):
        """Called before the backward pass of this rank's view ``slot``: stores the camera position in the row that
        will be gathered and returns the ``early_color`` argument of ``_C.rasterize_gaussians_backward_ex``."""
        P = self.P
        self.local[slot, 3 * P:3 * P + 3].copy_(cam.camera_center.reshape(3))
        if not self._on_gpu:
            return None
        self._early[slot] = False
        self._slot = slot
        return (self.local[slot, :3 * P], self._color_ready)

    def _color_ready(self):
        k = self._slot
        self._color_event[k].record(torch.cuda.current_stream(self.local.device))
        self._early[k] = True
        # over RCCL the gather leaves at once: it must be queued on every rank before the all-reduces of the runs
        if self._nccl():
            self._queue_gather(k, True)
            self._sent[k].add("gather")

    def begin_chunks(self, slot, dL_dmeans2D, radii, max_radii2D=None):
        self._slot = slot
        return self._plan(self._slot_chunks[slot], ("dL_dsh",), dL_dmeans2D, radii, max_radii2D, self._chunk_ready)

    def _nccl(self):
        return dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def _queue_gather(self, slot, early):
        self._mark("first", once=True)
        if early:
            # the gather waits for the hand-over event only, not for the per-Gaussian stage queued behind it
            self._side.wait_event(self._color_event[slot])
            with torch.cuda.stream(self._side):
                work = dist.all_gather_into_tensor(self.gathered[slot].view(-1), self.local[slot], group=self.group,
                                                   async_op=True)
        else:
            work = dist.all_gather_into_tensor(self.gathered[slot].view(-1), self.local[slot], group=self.group,
                                               async_op=True)
        self._pending.append(work)

    def _chunk_ready(self, k):
        """Run k of view ``self._slot`` is final once what is queued on the current stream has run: send it."""
        slot = self._slot
        if not self._nccl():
            return
        o, n = self._spans[k]
        ev = self._chunk_event[slot][k]
        ev.record(torch.cuda.current_stream(self.local.device))
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            self._pending.append(dist.all_reduce(self.per_view[slot, o:o + n], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
        self._sent[slot].add(k)

    def shaped(self):
        out = self._shaped_from(self._result_chunks)
        out["dL_dsh"] = self.dL_dsh
        return out

    def bytes_sent_per_rank(self):
        """Payload each rank contributes to the collectives of one step (float32 bytes)."""
        return 4 * self.V * (self.row + self.per_view.shape[1])

    def write_statistics(self, slot, dL_dmeans2D, radii, max_radii2D=None):
        v = self._slot_chunks[slot][0]
        _C.view_statistics(dL_dmeans2D, radii, v["stat_norm"], v["stat_vis"], max_radii2D)

    def post_view(self, slot, scene, cam, aux, group=None):
        """``aux`` is what ``forward_backward`` returned for this view (it carries the geometry buffer with the
        clamp flags and the raw colour gradient).  Queues whatever of the view's collectives -- one all-gather,
        one all-reduce per run of Gaussians -- has not left from inside the backward pass; with RCCL none is waited
        for here.  Every rank queues them in the same order: the gather, then the runs in order."""
        group = group if group is not None else self.group
        P = self.P
        early, self._early[slot] = self._early[slot], False
        sent, self._sent[slot] = self._sent[slot], set()
        if not early:  # the backward pass was not given begin_view()'s hand-over: derive the row now
            _C.masked_color_gradient(aux["geom"], aux["dL_dcolors"], out=self.local[slot, :3 * P])
            self.local[slot, 3 * P:3 * P + 3].copy_(cam.camera_center.reshape(3))
        if self._nccl():  # also with a single rank: same calls, trivial traffic
            if "gather" not in sent:
                self._queue_gather(slot, early)
            for k, (o, n) in enumerate(self._spans):
                if k not in sent:
                    self._pending.append(dist.all_reduce(self.per_view[slot, o:o + n], op=dist.ReduceOp.SUM,
                                                         group=group, async_op=True))
        elif self.world > 1:
            _all_gather_rows(self.gathered[slot], self.local[slot], group)
            _all_reduce_sum(self.per_view[slot], group)
        else:
            self.gathered[slot, 0].copy_(self.local[slot])

    def finish(self, scene, group=None, sh_adam=None):
        """Waits for the queued collectives, adds the views of the step up and rebuilds the summed SH gradient
        (views in (slot, rank) order: identical bits on every rank).  With ``sh_adam`` (see
        ``_C.sh_gradient_from_views``) that gradient is consumed by the coefficients' Adam step on the spot and
        ``shaped()['dL_dsh']`` is not updated."""
        P = self.P
        self._mark("own")
        for work in self._pending:
            work.wait()
        self._mark("done")
        self._clock_close()
        self._pending = []
        if self.V > 1:
            torch.sum(self.per_view, dim=0, out=self.reduced)
        rows = self.gathered.view(self.V * self.world, self.row)
        _C.sh_gradient_from_views(scene.means3D, rows[:, 3 * P:3 * P + 3], rows[:, :3 * P], scene.sh_degree, self.M,
                                  out=self.dL_dsh, sh_adam=sh_adam)
        return None

    def reduce(self, scene, cam, aux, group=None, sh_adam=None):
        self.post_view(0, scene, cam, aux, group)
        return self.finish(scene, group, sh_adam)


def sync_densification_stats(xyz_gradient_accum, denom, max_radii2D, group=None, sums=True):
    """Merge the per-rank densification statistics in place, right before they are used: the view-space gradient
    norms and the visibility counts (scene/gaussian_model.py:550-552) add up over the ranks' views, the largest
    screen radius (train.py:115) is the maximum.  Afterwards every rank holds the totals over all views seen since
    the last reset, so identical densification decisions follow; reset the statistics afterwards as the reference
    does (gaussian_model.py:455-457) -- merging twice without a reset would count the views twice.
    ``sums=False``: the two sums already travelled with the gradients of every step (``stats`` of the exchange
    classes) and every rank holds their totals; only the maximum, which may be merged at any time with the same
    result, is left to do."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if sums:
        both = torch.cat([xyz_gradient_accum.reshape(-1), denom.reshape(-1)])
        _all_reduce_sum(both, group)
        n = xyz_gradient_accum.numel()
        xyz_gradient_accum.copy_(both[:n].view_as(xyz_gradient_accum))
        denom.copy_(both[n:].view_as(denom))
    if max_radii2D.is_cuda and dist.get_backend(group) != "nccl":
        h = max_radii2D.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        max_radii2D.copy_(h)
    else:
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def shard_views(num_views, rank, world_size):
    """Indices of the views rank ``rank`` renders (round-robin)."""
    return list(range(rank, num_views, world_size))


def forward_backward(scene, cam, bg, dL_dpix_fn, bucket, tanfovx, tanfovy, scale_modifier=1.0, debug=False,
                     sh_adam=None, slot=0, max_radii2D=None):
    """One pass of the hot path for one view on this rank's GPU: rasterize,
    obtain dL/dpixel from ``dL_dpix_fn(color)``, run the backward pass with the
    parameter gradients landing in ``bucket.slot_views(slot)`` (either exchange form;
    ``bucket.post_view(slot, scene, cam, aux)`` then queues their exchange and ``bucket.finish(scene)`` completes
    the step; ``bucket.reduce(scene, cam, aux)`` does both for one view per rank).  ``sh_adam`` (see
    ``_C.rasterize_gaussians_backward_ex``) makes the backward pass apply the SH coefficients' optimiser step itself;
    single-rank single-view steps only.  A bucket built with ``stats`` also receives the view's densification
    statistics (and ``max_radii2D`` its running maximum).  Returns (color, radii, aux)."""
    e = torch.empty(0)
    H, W = int(cam.image_height), int(cam.image_width)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        bg, scene.means3D, e, scene.opacities, scene.scales, scene.rotations, scale_modifier, e,
        cam.world_view_transform, cam.full_proj_transform, tanfovx, tanfovy, H, W, scene.shs, scene.sh_degree,
        cam.camera_center, False, debug)
    dL_dpix = dL_dpix_fn(color)
    early = bucket.begin_view(cam, slot) if sh_adam is None and hasattr(bucket, "begin_view") else None
    if sh_adam is None and getattr(bucket, "chunked", False):
        # the per-Gaussian stage in runs of Gaussians, each run's gradients (and statistics) leaving for the other
        # ranks from inside the call
        dL_dmeans2D = torch.empty((scene.means3D.shape[0], 3), dtype=torch.float32, device=scene.means3D.device)
        plan = bucket.begin_chunks(slot, dL_dmeans2D, radii, max_radii2D)
        res = _C.rasterize_gaussians_backward_ex(
            bg, scene.means3D, radii, e, scene.scales, scene.rotations, scale_modifier, e, cam.world_view_transform,
            cam.full_proj_transform, tanfovx, tanfovy, dL_dpix, scene.shs, scene.sh_degree, cam.camera_center, geom, R,
            binning, img, debug, out={"dL_dmeans2D": dL_dmeans2D}, skip_sh=bucket.skip_sh, early_color=early,
            chunks=plan)
    else:
        out = {k: v for k, v in bucket.slot_views(slot).items() if k not in STAT_FIELDS}
        res = _C.rasterize_gaussians_backward_ex(
            bg, scene.means3D, radii, e, scene.scales, scene.rotations, scale_modifier, e, cam.world_view_transform,
            cam.full_proj_transform, tanfovx, tanfovy, dL_dpix, scene.shs, scene.sh_degree, cam.camera_center, geom, R,
            binning, img, debug, out=out, skip_sh=bucket.skip_sh, sh_adam=sh_adam, early_color=early)
        if getattr(bucket, "stats", False):
            bucket.write_statistics(slot, res[0], radii, max_radii2D)
    return color, radii, dict(num_rendered=R, dL_dmeans2D=res[0], dL_dcolors=res[1], dL_dcov3D=res[4], geom=geom)

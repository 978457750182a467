"""Multi-GPU sharding of the hemisphere frame: one process per GPU (torch.distributed over RCCL), no data-path
collective during the march, ONE gather of the finished bands to rank 0.

The unit of sharding is a band of `band_rows` pixel rows; rank r of N renders bands r, r+N, r+2N, ...
(interleaved, because cost per ray depends on cloud cover and elevation: contiguous eighths would leave the
zenith rank idle while the horizon ranks work).  It generalises the reference's update_position tile walk
(cloud_sky.gd:156-161): every band is rendered with the same frozen push-constant block (cloud_sky.gd:54-55,142)."""
import numpy as np

BAND_ROWS = 8  # one wavefront tile is 8x8 pixels; a band is one row of tiles


def bands_for_rank(height, rank, world, band_rows=BAND_ROWS):
    """(band_rows, first_band, band_stride, n_bands) for csky_render_clouds_device."""
    if height % band_rows:
        raise ValueError("frame height %d must be a multiple of the band height %d" % (height, band_rows))
    total = height // band_rows
    n = (total - rank + world - 1) // world if rank < total else 0
    return (band_rows, rank, world, n)


def max_bands(height, world, band_rows=BAND_ROWS):
    return (height // band_rows + world - 1) // world


def interleave(gathered, height, world, band_rows=BAND_ROWS):
    """gathered: [world, max_bands*band_rows, W, C] (rank-major compact bands, zero padded) -> [height, W, C]."""
    total = height // band_rows
    mb = max_bands(height, world, band_rows)
    xp = gathered
    g = xp.reshape(world, mb, band_rows, *xp.shape[2:])
    # band k of the frame = rank k % world, local band k // world
    if hasattr(g, "permute"):   # torch
        full = g.permute(1, 0, 2, *range(3, g.dim())).reshape(mb * world * band_rows, *xp.shape[2:])
        return full[: total * band_rows].contiguous()
    full = np.swapaxes(g, 0, 1).reshape(mb * world * band_rows, *xp.shape[2:])
    return np.ascontiguousarray(full[: total * band_rows])


def lut_rows_for_rank(lut_h, rank, world):
    """(first_row, row_stride, n_rows) for csky_render_sky_lut_rows_device: rank r of N renders rows r, r + N, ... of the sky LUT (sky_lut.gd:43-52
    renders it once per frame; N ranks rendering N identical copies would spend 33 us of a whole chip each, 12 % of a 1/8 frame share)."""
    return (rank, world, (lut_h - rank + world - 1) // world if rank < lut_h else 0)


def max_lut_rows(lut_h, world):
    return (lut_h + world - 1) // world


def render_sharded(render_bands, height, width, rank, world, dist=None, device=None, band_rows=BAND_ROWS):
    """Render this rank's bands and gather the frame on rank 0.

    render_bands(bands, out) fills `out` ([n_bands*band_rows, width, 4] uint16/int16 view of RGBA16F, torch tensor)
    for bands = (band_rows, first_band, band_stride, n_bands).  Returns the [height, width, 4] frame on rank 0,
    None elsewhere.  `dist` = torch.distributed (None or world == 1: no collective)."""
    import torch

    mb = max_bands(height, world, band_rows)
    local = torch.zeros((mb * band_rows, width, 4), dtype=torch.int16, device=device)
    b = bands_for_rank(height, rank, world, band_rows)
    render_bands(b, local[: b[3] * band_rows])
    if world == 1 or dist is None:
        return local[:height]
    # the collective moves raw bytes: neither RCCL nor gloo has a 16-bit integer type, uint8 works on both
    lb = local.view(torch.uint8)
    if rank == 0:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(lb, gather_list=[gathered[i].view(torch.uint8) for i in range(world)], dst=0)
        return interleave(gathered, height, world, band_rows)
    dist.gather(lb, gather_list=None, dst=0)
    return None


class FrameGroups:
    """Frame groups for throughput workloads (BASELINE config 5's 64-frame sweep): the `world` ranks are split into `groups` groups of
    world / groups; frame f belongs to group f % groups, whose ranks split ITS bands (world / groups)-way; every frame is gathered on rank 0.
    Group g's collective runs on a communicator of {0} + its ranks: rank 0 takes part in every frame's gather and contributes an unused
    dummy when it is not a member.  groups = 1 is the plain split (every rank works on every frame, the world communicator).
    The same dealing as csky_multi_set_groups behind the C ABI."""

    def __init__(self, rank, world, groups=1, dist=None):
        if groups < 1 or world % groups:
            raise ValueError("the group count %d must divide the world size %d" % (groups, world))
        self.rank, self.world, self.groups, self.dist = int(rank), int(world), int(groups), dist
        self.per = world // groups
        self.my_group, self.index = rank // self.per, rank % self.per
        self.members = [sorted(set(([0] if groups > 1 else []) + list(range(g * self.per, (g + 1) * self.per)))) for g in range(groups)]
        self.pgs = [None] * groups                       # None = the world communicator
        if dist is not None and world > 1 and groups > 1:
            for g in range(groups):                      # every rank creates every group, in the same order (a collective)
                self.pgs[g] = dist.new_group(self.members[g])

    @property
    def max_members(self):
        return max(len(m) for m in self.members)

    def bands(self, height, band_rows=BAND_ROWS):
        return bands_for_rank(height, self.index, self.per, band_rows)

    def max_bands(self, height, band_rows=BAND_ROWS):
        return max_bands(height, self.per, band_rows)

    def lut_rows(self, lut_h):
        return lut_rows_for_rank(lut_h, self.index, self.per)

    def max_lut_rows(self, lut_h):
        return max_lut_rows(lut_h, self.per)

    def group_of(self, frame_no):
        return frame_no % self.groups

    def renders(self, frame_no):
        return self.group_of(frame_no) == self.my_group

    def takes_part(self, frame_no):
        return self.renders(frame_no) or self.rank == 0

    def gather(self, frame_no, src_bytes, gathered=None, async_op=False):
        """src_bytes: this rank's compact bands as a uint8 view (any content when rank 0 is not a member); gathered (rank 0): a tensor of at
        least len(members) band buffers.  Returns the collective's work handle (async_op) or None."""
        g = self.group_of(frame_no)
        glist = None
        if self.rank == 0:
            import torch
            glist = [gathered[i].view(torch.uint8) for i in range(len(self.members[g]))]
        return self.dist.gather(src_bytes, gather_list=glist, dst=0, group=self.pgs[g], async_op=async_op)

    def assemble(self, frame_no, gathered, height, band_rows=BAND_ROWS):
        """Rank 0, after the gather of frame_no completed: [height, W, C]."""
        g = self.group_of(frame_no)
        got = gathered[: len(self.members[g])]
        if self.groups > 1 and g != 0:
            got = got[1:]                                # rank 0's dummy contribution to another group's gather
        return interleave(got, height, self.per, band_rows)

    def split(self, gathered_bytes, height, width, lut_h=0, lut_w=0, band_rows=BAND_ROWS):
        """Rank 0: a gathered [members, bytes] uint8 tensor whose rows are (compact bands | compact sky-LUT rows) -> the two typed tensors
        ([members, max_bands*band_rows, W, 4], [members, max_lut_rows, lut_w, 4] int16) that assemble() / assemble_lut() take."""
        import torch
        bb = self.max_bands(height, band_rows) * band_rows * width * 8
        m = gathered_bytes.shape[0]
        img = gathered_bytes[:, :bb].view(torch.int16).reshape(m, -1, width, 4)
        lut = gathered_bytes[:, bb: bb + self.max_lut_rows(lut_h) * lut_w * 8].view(torch.int16).reshape(m, -1, lut_w, 4) if lut_h else None
        return img, lut

    def assemble_device(self, frame_no, gathered_bytes, ctx, stream, height, width, out_frame, lut_h=0, lut_w=0, out_lut=None, band_rows=BAND_ROWS):
        """Rank 0, device tensors: the same as split() + assemble() (+ assemble_lut()) with the library's narrow copy kernel
        (csky_interleave_bands_device) on `stream` (a raw HIP stream handle) into the caller's out_frame [height, W, 4] int16 (and out_lut):
        an HBM-bound pass that runs beside the following frames' marches instead of sweeping the whole chip."""
        g = self.group_of(frame_no)
        skip = 1 if (self.groups > 1 and g != 0) else 0          # rank 0's dummy contribution to another group's gather
        stride = gathered_bytes.stride(0) * gathered_bytes.element_size()
        base = gathered_bytes.data_ptr() + skip * stride
        bb = self.max_bands(height, band_rows) * band_rows * width * 8
        ctx.interleave_bands_device(base, stride, self.per, band_rows * width * 8, height // band_rows, out_frame.data_ptr(), stream)
        if lut_h:
            ctx.interleave_bands_device(base + bb, stride, self.per, lut_w * 8, lut_h, out_lut.data_ptr(), stream)

    def assemble_lut(self, frame_no, gathered_lut, lut_h):
        """Rank 0: the members' sky-LUT rows of frame_no ([members, max_lut_rows, w, 4]) -> the [lut_h, w, 4] LUT."""
        return self.assemble(frame_no, gathered_lut, lut_h, band_rows=1)

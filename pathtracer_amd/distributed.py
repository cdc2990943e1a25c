"""Sample-sharded rendering across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI), the scene replicated, FrameCount values dealt round-robin.

Every (pixel, FrameCount) sample is independent (the shader seeds its RNG with
pixelIndex + FrameCount * 719393, Shaders/pathtracing.comp:213-214), so rank r of G renders
FrameCount r+1, r+1+G, r+1+2G, ... over the full frame and the union over ranks is exactly the
single-GPU sample set.  Ranks accumulate SUMS (PT_ACCUM_SUM) into torch-owned device buffers that
the kernel writes directly (pt_bind_images); the only data-path collective is one sum all-reduce of
those buffers per render, after which every rank holds the total and divides by the sample count.
The reference has no multi-GPU path; this is new.
"""
import torch

from . import renderer as R


def shard_frames(rank, world, total_spp, first_frame=1):
    """(first FrameCount, count, stride) of rank `rank` when `total_spp` consecutive FrameCount
    values starting at `first_frame` are dealt round-robin to `world` ranks."""
    count = (total_spp - rank + world - 1) // world if total_spp > rank else 0
    return first_frame + rank, count, world


def reduce_sum(images, group=None):
    """In-place sum over ranks of a stacked accumulation tensor (the path's only collective).  Device
    tensors go to RCCL as they are; under a gloo group (CPU rehearsals of the N > 1 path, ranks that
    share a GPU) a device tensor is staged through host memory, so that gloo only ever sees host
    tensors."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if images.is_cuda and dist.get_backend(group) == "gloo":
            host = images.cpu()  # synchronises the current stream
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            images.copy_(host)
        else:
            dist.all_reduce(images, op=dist.ReduceOp.SUM, group=group)
    return images


class ShardedRenderer:
    def __init__(self, width, height, device_index, rank=0, world=1):
        self.rank, self.world = int(rank), int(world)
        self.width, self.height = width, height
        self.device = torch.device("cuda", device_index)
        self.renderer = R.Renderer(width, height, device=device_index)
        # Kernels, clears and collectives are all ordered on one torch-owned stream (torch's default
        # stream is the null stream, which the C ABI reads as "use the context's own stream").
        self.stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self.stream):
            # one flat buffer for the three images -> one collective
            self.images = torch.zeros(3, height, width, 4, device=self.device, dtype=torch.float32)
            self.total = torch.zeros_like(self.images) if self.world > 1 else self.images
        self.stream.synchronize()
        self.renderer.bind_images(*(self.images[k].data_ptr() for k in range(3)))
        self.renderer.set_stream(self.stream.cuda_stream)
        self.samples_local = 0
        # The collective runs on its own stream so that the next step's kernel does not wait for it.
        self.comm_stream = torch.cuda.Stream(self.device) if self.world > 1 else None
        self._reduced = None  # event: the last all-reduce into `total` has finished

    def set_params(self, params):
        params.accumulate_mode = 1 if self.world > 1 else params.accumulate_mode
        self._sum_mode = params.accumulate_mode == 1
        self.renderer.set_params(params)

    def clear(self):
        with torch.cuda.stream(self.stream):
            self.images.zero_()
        self.samples_local = 0

    def frames_of_step(self, step, spp):
        """First FrameCount, count and stride of this rank's `spp` samples in step `step`."""
        first = 1 + self.rank + step * spp * self.world
        return first, spp, self.world

    def render_step(self, step, spp, counted=False):
        first, count, stride = self.frames_of_step(step, spp)
        self.renderer.dispatch(first, count, stride, counted=counted)
        self.samples_local += count

    def reduce(self):
        """Sum the accumulation buffers over ranks into `self.total` (one RCCL all-reduce of the
        three images); this rank's own sums stay intact so rendering can continue: the snapshot is
        taken on the render stream, the collective runs on a second stream and overlaps the next
        step's kernel.  A no-op on one GPU, where `total` aliases `images`."""
        if self.world > 1:
            with torch.cuda.stream(self.stream):
                if self._reduced is not None:
                    self.stream.wait_event(self._reduced)  # the previous collective still reads `total`
                self.total.copy_(self.images)
                copied = torch.cuda.Event()
                copied.record(self.stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(copied)
                reduce_sum(self.total)
                self._reduced = torch.cuda.Event()
                self._reduced.record(self.comm_stream)

    def resolve(self, total_samples):
        """Mean images after reduce(): the summed buffers divided by the global sample count."""
        with torch.cuda.stream(self.stream):
            if self._reduced is not None:
                self.stream.wait_event(self._reduced)
            out = self.total / float(total_samples) if self._sum_mode else self.total.clone()
        self.stream.synchronize()
        return out

"""Model protocol of the hot path: the methods LogTrainer calls on netG / netD
(diagan-pkg/diagan/trainer/trainer.py:257-291; contract listed in SURVEY §8(b)).

These mirror torch_mimicry's BaseGenerator / BaseDiscriminator / BaseModel (the reference imports
them at diagan-pkg/diagan/models/mnist.py:4 and through torch_mimicry.nets.sngan): same method
names, argument names and return values; the bodies run on the HIP engine with an explicit
backward instead of autograd.
"""
import os

import torch

from diagan.models.layers import FlatNet
from diagan.ops import eltwise as E


PREFETCH_FAKES = os.environ.get("DIAGAN_PREFETCH_FAKES", "1") != "0"
# the generator update's own forward (with its backward context) rides as the LAST batch of the same stacked forward
# (+1.8 % on the SNGAN-32 step: its 8x8 / 16x16 launches at batch 64 run at low occupancy on their own); 0 = separate
STACK_G_STEP = os.environ.get("DIAGAN_STACK_G_STEP", "1") != "0"


def _world_size():
    from diagan.trainer import distributed as dist
    return dist.get_world_size()


class BaseModel(FlatNet):
    """Checkpoint I/O with mimicry's file layout: {model_state_dict, optimizer_state_dict,
    global_step} at <directory>/<basename(directory)>_<step>_steps.pth (consumers:
    train_mimicry_phase1.py:97-98, train_mimicry_phase2.py:98-101)."""

    def restore_checkpoint(self, ckpt_file, optimizer=None):
        if not ckpt_file:
            raise ValueError("No checkpoint file to be restored.")
        # always through host memory: rank 0 saved device tensors of cuda:0, and under torch.distributed.run every rank
        # would otherwise open a context on GPU 0 just to deserialise; load_state_dict copies into the flat slabs anyway
        ckpt_dict = torch.load(ckpt_file, map_location='cpu', weights_only=False)
        self.load_state_dict(ckpt_dict['model_state_dict'])
        self.param_version += 1
        if optimizer:
            optimizer.load_state_dict(ckpt_dict['optimizer_state_dict'])
        return ckpt_dict['global_step']

    def save_checkpoint(self, directory, global_step, optimizer=None, name=None):
        if not os.path.exists(directory):
            os.makedirs(directory)
        ckpt_dict = {
            'model_state_dict': self.state_dict(),
            'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else None,
            'global_step': global_step,
        }
        if name is None:
            name = "{}_{}_steps.pth".format(os.path.basename(directory), global_step)
        torch.save(ckpt_dict, os.path.join(directory, name))

    def count_params(self):
        return FlatNet.count_params(self)


class BaseGenerator(BaseModel):
    def __init__(self, nz, ngf, bottom_width, loss_type, **kwargs):
        super().__init__()
        self.nz, self.ngf, self.bottom_width, self.loss_type = nz, ngf, bottom_width, loss_type

    # -- subclasses implement: forward_nhwc(z, training, save) -> (img NHWC4, ctx); backward_nhwc(ctx, g)
    def forward(self, x):
        """noise [n, nz] -> images NCHW [n, 3, H, W] (reference-visible tensor layout)."""
        img, _ = self.forward_nhwc(x, self.training, save=False)
        return E.nhwc_to_nchw(img, self.out_channels)

    def generate_images(self, num_images, device=None, noise=None):
        if device is None:
            device = self.device
        if noise is None:
            noise = torch.randn((num_images, self.nz), device=device)
        return self.forward(noise)

    def generate_images_nhwc(self, num_images, device=None, noise=None, save=False, out=None):
        if device is None:
            device = self.device
        if noise is None and not save:
            ready = self._take_prefetched(num_images)
            if ready is not None:
                if out is not None:
                    out.copy_(ready)
                    ready = out
                return ready, None
        if noise is None and save and out is None:
            ready = getattr(self, '_g_step_ready', None)
            self._g_step_ready = None
            if ready is not None and ready[0].shape[0] == num_images and ready[2] == self.param_version and self.training:
                return ready[0], ready[1]
        if noise is None:
            noise = torch.randn((num_images, self.nz), device=device)
        return self.forward_nhwc(noise, self.training, save=save, out=out)

    # ---- fake batches of a whole global step in ONE forward --------------------------------------------------
    # The n_dis discriminator updates of a global step (trainer.py:250-277) each start with G(z) under no_grad, and G
    # does not change until its own update at the end of the step: the forwards are independent.  `prefetch_fakes`
    # draws the noise batches in the order the updates would (same generator state afterwards) and runs them as one
    # stacked forward with per-batch BatchNorm statistics; `generate_images_nhwc` then hands the batches out in order.
    supports_stacked_forward = False
    max_stacked_images = 0        # largest stacked batch whose activations stay below 2 GiB

    def prefetch_fakes(self, count, batch_size, device=None, g_step=False):
        """g_step: the global step ends with this generator's own update on a batch of the same size (trainer.py:279-284 of
        the reference).  Its noise is then drawn right after the discriminator updates' -- the order in which the updates
        would draw them: nothing else consumes the device generator in between for the models that stack -- and its
        forward, WITH the context for its backward, is the last batch of the stacked forward;
        generate_images_nhwc(save=True) hands it out (DIAGAN_STACK_G_STEP=0: separate forward as before)."""
        self._fake_pool, self._g_step_ready = [], None
        if not (self.supports_stacked_forward and self.training and count > 1 and PREFETCH_FAKES):
            return
        device = self.device if device is None else device
        # every activation of the stacked forward must stay below 2 GiB (32-bit buffer offsets in the kernels)
        per_chunk = min(count, self.max_stacked_images // max(batch_size, 1))
        if per_chunk < 2:
            return
        if g_step and STACK_G_STEP and (count + 1) * batch_size <= self.max_stacked_images:
            noise = [torch.randn((batch_size, self.nz), device=device) for _ in range(count + 1)]   # in update order
            imgs, ctx = self.forward_nhwc(torch.cat(noise), True, save=True, groups=count + 1, save_group=count)
            parts = imgs.split(batch_size)
            self._fake_pool.extend(parts[:count])
            self._fake_pool_version = self.param_version
            self._g_step_ready = (parts[count], ctx, self.param_version)
            return
        noise = [torch.randn((batch_size, self.nz), device=device) for _ in range(count)]   # in update order
        for lo in range(0, count, per_chunk):
            part = noise[lo: lo + per_chunk]
            imgs, _ = self.forward_nhwc(torch.cat(part), True, save=False, groups=len(part))
            self._fake_pool.extend(imgs.split(batch_size))
        self._fake_pool_version = self.param_version

    def _take_prefetched(self, num_images):
        pool = getattr(self, '_fake_pool', None)
        if not pool:
            return None
        if pool[0].shape[0] != num_images or self._fake_pool_version != self.param_version or not self.training:
            self._fake_pool = []        # changed parameters, eval mode, or a caller that asks for another batch size
                                        # (LogTrainer does not prefetch a step that contains a ragged batch): fall back
            return None
        return pool.pop(0)

    def compute_gan_loss(self, output):
        k = None
        if getattr(self, 'use_topk', False):
            k = int(self.topk_rate * output.shape[0])
        errG, _ = E.loss_gen(output.detach().contiguous(), self.loss_type, k=k, need_grad=False)
        return errG[0]

    def train_step(self, real_batch, netD, optG, log_data, device=None, global_step=None, scaler=None, noise=None,
                   **kwargs):
        """One G update (restated base step, topk_models.py:46-116 / mnist.py:82-152)."""
        if scaler is not None:
            raise NotImplementedError("amp/GradScaler is not part of the fp32 MI355X path")
        self.zero_grad()
        self.wgrad_batch.hold = _world_size() > 1      # the weight-gradient reduction overlaps the exchange (sync_grads)
        batch_size = real_batch[0].shape[0]
        fake, gctx = self.generate_images_nhwc(batch_size, device=device, noise=noise, save=True)
        output, dctx = netD.forward_nhwc(fake, netD.training, save=True, need_dgrad=True, need_in_dgrad=True)
        k = None
        if getattr(self, 'use_topk', False):          # get_topk, topk_models.py:31-38
            k = int(self.topk_rate * batch_size)
        errG, dlogit = E.loss_gen(output, self.loss_type, k=k)
        # backward: through D to the images (D's own weight gradients are dead values in the
        # reference -- zeroed at the next D step before use -- so they are not computed), then G
        g_img = netD.backward_nhwc(dctx, dlogit, need_wgrad=False, need_gx=True)
        self.backward_nhwc(gctx, g_img)
        self.sync_grads(optG)
        optG.step()
        log_data.add_metric('errG', errG[0], group='loss')
        return log_data


class BaseDiscriminator(BaseModel):
    supports_defer_step = True      # train_step(defer_step=True) leaves optD.step() to the caller (LogTrainer, DP phase 2)

    def __init__(self, ndf, loss_type, **kwargs):
        super().__init__()
        self.ndf, self.loss_type = ndf, loss_type
        self.use_gold = False

    # -- subclasses implement forward_nhwc(x NHWC4, training, save, need_dgrad) -> (logit [n,1], ctx)
    #    and backward_nhwc(ctx, dlogit, need_wgrad, need_gx) -> g_x or None
    def to_nhwc(self, x):
        return E.nchw_to_nhwc(x.to(dtype=torch.float32), self.in_channels_padded)

    def forward(self, x):
        """images NCHW [n, C, H, W] -> logits [n, 1]."""
        logit, _ = self.forward_nhwc(self.to_nhwc(x), self.training, save=False, need_dgrad=False)
        return logit

    def compute_gan_loss(self, output_real, output_fake):
        out3, _, _ = E.loss_dis(output_real.detach().contiguous(), output_fake.detach().contiguous(),
                                self.loss_type, gold=self.use_gold, need_grad=False)
        return out3[0]

    def compute_probs(self, output_real, output_fake):
        out3, _, _ = E.loss_dis(output_real.detach().contiguous(), output_fake.detach().contiguous(),
                                self.loss_type, gold=False, need_grad=False)
        return out3[1].item(), out3[2].item()

    def train_step(self, real_batch, netG, optD, log_data, device=None, global_step=None, scaler=None, noise=None,
                   **kwargs):
        """One D update (torch_mimicry BaseDiscriminator.train_step; call site trainer.py:257-264)."""
        if scaler is not None:
            raise NotImplementedError("amp/GradScaler is not part of the fp32 MI355X path")
        self.zero_grad()
        self.wgrad_batch.hold = _world_size() > 1      # the weight-gradient reduction overlaps the exchange (sync_grads)
        real_images = real_batch[0]
        batch_size = real_images.shape[0]
        if getattr(self, 'pair_forward', False) and self.training:
            # D(real) and D(fake) as ONE batched pass: the two forwards differ only by their spectral-norm
            # sigma (u is still advanced twice), which is applied per half of the batch in the GEMM epilogue.
            B, _, H, W = real_images.shape
            x_all = torch.empty((2 * B, H, W, self.in_channels_padded), dtype=torch.float32, device=real_images.device)
            E.nchw_to_nhwc(real_images.to(dtype=torch.float32), self.in_channels_padded, out=x_all[:B])
            netG.generate_images_nhwc(batch_size, device=device, noise=noise, save=False, out=x_all[B:])   # .detach()
            out_all, ctx = self.forward_nhwc(x_all, True, save=True, need_dgrad=True, need_in_dgrad=False, slot='pair')
            dl = torch.empty(2 * B, dtype=torch.float32, device=x_all.device)
            out3, _, _ = E.loss_dis(out_all[:B], out_all[B:], self.loss_type, gold=self.use_gold, d_real=dl[:B],
                                    d_fake=dl[B:])
            self.backward_nhwc(ctx, dl, need_wgrad=True, need_gx=False)
        else:
            out_real, ctx_r = self.forward_nhwc(self.to_nhwc(real_images), self.training, save=True, need_dgrad=True,
                                                need_in_dgrad=False)
            fake, _ = netG.generate_images_nhwc(batch_size, device=device, noise=noise, save=False)   # .detach()
            out_fake, ctx_f = self.forward_nhwc(fake, self.training, save=True, need_dgrad=True, need_in_dgrad=False,
                                                slot=1)
            out3, d_real, d_fake = E.loss_dis(out_real, out_fake, self.loss_type, gold=self.use_gold)
            self.backward_nhwc(ctx_r, d_real, need_wgrad=True, need_gx=False)
            self.backward_nhwc(ctx_f, d_fake, need_wgrad=True, need_gx=False)
        # defer_step (LogTrainer, phase 2 under data parallelism): the all-reduce of this update stays in flight and
        # the caller runs optD.step() after the NEXT network's update has been issued (D and D_drs are independent,
        # diagan-pkg/diagan/trainer/trainer.py:250-277)
        defer = bool(kwargs.get('defer_step', False))
        self.sync_grads(optD, async_op=defer)
        if not defer:
            optD.step()
        # device scalars: no host sync here (the reference calls .item() three times per D step)
        log_data.add_metric('errD', out3[0], group='loss')
        log_data.add_metric('D(x)', out3[1], group='prob')
        log_data.add_metric('D(G(z))', out3[2], group='prob')
        return log_data

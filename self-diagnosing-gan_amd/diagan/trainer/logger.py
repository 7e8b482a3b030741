"""Minimal stand-ins for torch_mimicry.training.{metric_log.MetricLog, logger.Logger}, the two
helper objects LogTrainer uses (diagan-pkg/diagan/trainer/trainer.py:113-117,239,304-326).

Metric values may be 0-dim device tensors; they are only converted to Python floats when a line
is actually printed / written, so logging never forces a host sync inside the hot loop."""
import math
import os


class Metric:
    def __init__(self, value, group, precision):
        self._value, self.group, self.precision = value, group, precision

    @property
    def value(self):
        v = self._value
        if hasattr(v, 'item'):
            v = v.item()
            self._value = v
        return v


class MetricLog:
    def __init__(self, **kwargs):
        self.metrics_dict = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.metrics_dict[name] = Metric(value, group, precision)

    def items(self):
        return self.metrics_dict.items()

    def __getitem__(self, name):
        return round(self.metrics_dict[name].value, self.metrics_dict[name].precision)

    def get_group_name(self, name):
        return self.metrics_dict[name].group


class Logger:
    def __init__(self, log_dir, num_steps, dataset_size, device=None, flush_secs=120, writers=None, **kwargs):
        self.log_dir, self.num_steps, self.dataset_size = log_dir, num_steps, dataset_size
        self.num_epochs = int(math.ceil(num_steps / max(dataset_size, 1)))
        self.lines = []
        self.quiet = bool(int(os.environ.get("DIAGAN_QUIET", "0")))

    def write_summaries(self, log_data, global_step):
        """Scalars go to <log_dir>/scalars.csv (tensorboard is not a dependency of the hot path)."""
        path = os.path.join(self.log_dir, "scalars.csv")
        with open(path, "a") as f:
            for name, m in log_data.items():
                f.write(f"{global_step},{name},{m.value}\n")

    def print_log(self, global_step, log_data, time_taken):
        parts = [f"INFO: [Epoch {max(global_step - 1, 0) // max(self.dataset_size, 1) + 1:d}/{self.num_epochs:d}]"
                 f"[Global Step: {global_step:d}/{self.num_steps:d}]"]
        vals = [f"{name}: {m.value:.{m.precision}f}" if isinstance(m.value, float) else f"{name}: {m.value}"
                for name, m in sorted(log_data.items()) if not name.startswith('lr_')]
        line = parts[0] + " \n| " + "\n| ".join(vals) + f"\n| ({time_taken:.4f} sec/idx)"
        if not self.quiet:
            print(line)
        return line

    def vis_images(self, netG, global_step, num_images=64):
        """Fixed-noise sample grid; written as a raw tensor (torchvision is not available)."""
        import torch
        img_dir = os.path.join(self.log_dir, 'images')
        os.makedirs(img_dir, exist_ok=True)
        gen = torch.Generator(device='cpu').manual_seed(0)
        noise = torch.randn((num_images, netG.nz), generator=gen).to(netG.device)
        was_training = netG.training
        netG.eval()
        images = netG.generate_images(num_images, noise=noise).detach().cpu()
        netG.train(was_training)
        torch.save(images, os.path.join(img_dir, f'fixed_fake_samples_step_{global_step}.pt'))

    def close_writers(self):
        pass

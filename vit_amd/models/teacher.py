"""The CIFAR ResNet-20 / -32 teacher of the distillation configuration, on stock PyTorch-ROCm modules (MIOpen): it runs
frozen, in eval mode and under no_grad, and only its logits enter the hot path (the fused distillation loss), so it is not
part of the kernel library.  Architecture and parameter names follow the reference's teacher (models/resnet.py:21-85 there:
3 stages of 3x3 basic blocks at 16 / 32 / 64 channels, parameter-free "option A" shortcuts, global average pool, `fc`),
so the bare state_dict of its checkpoints/resnet20_cifar10.pth loads with strict=True (load it with
torch.load(..., weights_only=True))."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
	def __init__(self, cin, cout, stride):
		super().__init__()
		self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
		self.bn1 = nn.BatchNorm2d(cout)
		self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
		self.bn2 = nn.BatchNorm2d(cout)
		self.pad = (cout - cin) // 2 if (stride != 1 or cin != cout) else None
		self.stride = stride

	def forward(self, x):
		y = F.relu(self.bn1(self.conv1(x)))
		y = self.bn2(self.conv2(y))
		if self.pad is not None:  # option A: subsample, zero-pad the new channels
			x = F.pad(x[:, :, ::self.stride, ::self.stride], (0, 0, 0, 0, self.pad, self.pad))
		return F.relu(y + x)


class CifarResNet(nn.Module):
	def __init__(self, depth=20, num_classes=10):
		super().__init__()
		if depth not in (20, 32):
			raise ValueError(f"Invalid ResNet depth for CIFAR: {depth}")
		n = (depth - 2) // 6
		self.conv1 = nn.Conv2d(3, 16, 3, 1, 1, bias=False)
		self.bn1 = nn.BatchNorm2d(16)
		cin, stages = 16, []
		for cout, stride in ((16, 1), (32, 2), (64, 2)):
			blocks = []
			for i in range(n):
				blocks.append(_Block(cin, cout, stride if i == 0 else 1))
				cin = cout
			stages.append(nn.Sequential(*blocks))
		self.layer1, self.layer2, self.layer3 = stages
		self.fc = nn.Linear(64, num_classes)
		for m in self.modules():
			if isinstance(m, (nn.Conv2d, nn.Linear)):
				nn.init.kaiming_normal_(m.weight)

	def forward(self, x):
		x = F.relu(self.bn1(self.conv1(x)))
		x = self.layer3(self.layer2(self.layer1(x)))
		return self.fc(x.mean((2, 3)))


def fold_batchnorms(model):
	"""In place, for a teacher that stays in eval mode: every BatchNorm2d behind a bias-free convolution becomes part of
	that convolution (w' = w * gamma / sqrt(var + eps) per output channel, b' = beta - mean * gamma / sqrt(var + eps), formed
	in float64 and stored as float32), and the BatchNorm module is replaced by nn.Identity.  The function computed is the
	same up to float32 rounding; the per-step BatchNorm kernels (a third of the distillation step's time at batch 512) are
	gone.  After folding the module no longer has the reference's state_dict keys: load first, then fold."""
	def fold(owner, conv_name, bn_name):
		conv, bn = getattr(owner, conv_name), getattr(owner, bn_name)
		assert isinstance(bn, nn.BatchNorm2d) and conv.bias is None and not bn.training
		scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
		w = (conv.weight.double() * scale.view(-1, 1, 1, 1)).float()
		b = (bn.bias.double() - bn.running_mean.double() * scale).float()
		new = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True).to(w.device)
		new.weight = nn.Parameter(w, requires_grad=False)
		new.bias = nn.Parameter(b, requires_grad=False)
		setattr(owner, conv_name, new)
		setattr(owner, bn_name, nn.Identity())
	with torch.no_grad():
		fold(model, "conv1", "bn1")
		for stage in (model.layer1, model.layer2, model.layer3):
			for blk in stage:
				fold(blk, "conv1", "bn1")
				fold(blk, "conv2", "bn2")
	return model


def frozen_teacher(depth=20, num_classes=10, state_dict=None, fold_bn=False):
	"""Teacher as DeiTTrainer holds it (deit_trainer.py:15-18 there): eval mode, no gradients.  fold_bn: fold the (frozen)
	BatchNorms into the convolutions in front of them (fold_batchnorms)."""
	m = CifarResNet(depth, num_classes)
	if state_dict is not None:
		m.load_state_dict(state_dict, strict=True)
	m.eval()
	for p in m.parameters():
		p.requires_grad_(False)
	if fold_bn:
		fold_batchnorms(m)
	return m


class TeacherBeside:
	"""The frozen teacher's forward on a stream of its own, beside whatever the caller queues next (the student's forward): the
	two are independent until the distillation loss (deit_trainer.py:33-45 of the reference runs them one after the other), and
	both are chains of short kernels that leave most of the chip idle.  `start(x)` queues the teacher behind everything
	already queued on the current stream; `logits()` makes the current stream wait for it and returns its output.  Under HIP
	graph capture (and on the CPU) the teacher simply runs in place."""

	def __init__(self, teacher):
		self.teacher = teacher
		self.stream = None
		self.out = None
		self.beside = False

	def start(self, x):
		self.beside = x.is_cuda and not torch.cuda.is_current_stream_capturing()
		if not self.beside:
			with torch.no_grad():
				self.out = self.teacher(x)
			return
		if self.stream is None or self.stream.device != x.device:
			self.stream = torch.cuda.Stream(device=x.device)
		main = torch.cuda.current_stream(x.device)
		self.stream.wait_stream(main)
		with torch.cuda.stream(self.stream), torch.no_grad():
			self.out = self.teacher(x)
		self.out.record_stream(main)  # (allocated on the side stream, read on the main one)
		x.record_stream(self.stream)

	def logits(self):
		out, self.out = self.out, None
		if self.beside:
			torch.cuda.current_stream(out.device).wait_stream(self.stream)
		return out

"""Inference-side host API (work in progress this round): generation configs and tasks behind the reference's infer.py surface."""
from __future__ import annotations

import dataclasses
import itertools
import re
from typing import Any, Optional, Sequence

import torch


def format_semifix(value: float, precision: int = 3) -> str:
	"""Compact number formatting used in generation-config names (reference utils.format_semifix): fixed notation, trailing zeros stripped."""
	s = f"{value:.{precision}f}".rstrip("0").rstrip(".")
	return s if s not in ("", "-0") else "0"


@dataclasses.dataclass(frozen=True)
class GenerationConfig:
	"""reference infer.py:357-433"""
	method: str
	topk: int
	vocab_prior: bool
	vocab_per_token: bool
	vocab_scaler: float
	guided: bool
	guide_renorm: bool
	temperature: float
	length_alpha: float
	name: str = dataclasses.field(init=False)

	def __post_init__(self):
		object.__setattr__(self, "name", self.generate_name())

	def generate_name(self) -> str:
		prior = f"{'tok' if self.vocab_per_token else 'tgt'}{format_semifix(self.vocab_scaler)}" if self.vocab_prior else "none"
		guide = "n" if not self.guided else ("r" if self.guide_renorm else "p")
		return f"{self.method}_k{self.topk}_v{prior}_g{guide}_t{format_semifix(self.temperature)}_a{format_semifix(self.length_alpha)}"

	@staticmethod
	def from_name(name: str) -> "GenerationConfig":
		parts = name.split("_")
		kw = dict(method=parts[0], topk=0, vocab_prior=False, vocab_per_token=False, vocab_scaler=0.0, guided=False, guide_renorm=False, temperature=1.0, length_alpha=0.0)
		for part in parts[1:]:
			if not part:
				raise ValueError(f"Unexpected multiple underscores in generation configuration: {name}")
			key, val = part[:1], part[1:]
			try:
				if key == "k":
					kw["topk"] = int(val)
				elif key == "v":
					if val != "none":
						m = re.fullmatch(r"(tok|tgt)(.*)", val)
						if not m:
							raise ValueError(f"Invalid vocab prior specification: {val}")
						kw.update(vocab_prior=True, vocab_per_token=(m.group(1) == "tok"), vocab_scaler=float(m.group(2)))
				elif key == "g":
					if val not in ("n", "p", "r"):
						raise ValueError(f"Invalid guide specification: {val}")
					kw.update(guided=(val != "n"), guide_renorm=(val == "r"))
				elif key == "t":
					kw["temperature"] = float(val)
				elif key == "a":
					kw["length_alpha"] = float(val)
				else:
					raise ValueError(f"Invalid prefix: {key}")
			except ValueError:
				raise ValueError(f"Failed to parse generation configuration part: {part}")
		cfg = GenerationConfig(**kw)
		if cfg.method not in ("greedy", "beam", "all"):
			raise ValueError(f"Invalid generation configuration method: {cfg.method}")
		if cfg.topk < 1:
			raise ValueError(f"Missing or invalid non-positive generation configuration top-k: {cfg.topk}")
		if cfg.temperature <= 0:
			raise ValueError(f"Invalid non-positive generation configuration temperature tau: {cfg.temperature}")
		assert cfg.name == name
		return cfg


# ------------------------------------------------------------------------------------------------------------------------------
# generation task (reference infer.py:436-644): dispatch to the decoder's generate* + host-side validity / top-k bookkeeping
# ------------------------------------------------------------------------------------------------------------------------------

import contextlib  # noqa: E402
import enum  # noqa: E402
import gc  # noqa: E402
import math  # noqa: E402
import os  # noqa: E402
from typing import Callable, Iterable, Iterator, Type, Union  # noqa: E402

from . import embedders, embedding_dataset, embedding_decoder, utils  # noqa: E402


class PredictionType(enum.Enum):
	Correct = 0
	ValidGuide = 1
	ValidVocab = 2
	Other = 3


@dataclasses.dataclass(frozen=True)
class NOVICOutput:
	embeds: torch.Tensor
	preds: tuple
	logprobs: tuple
	probs: tuple
	types: tuple


@dataclasses.dataclass(eq=False)
class GenerationTask:
	COLOR_MAP = ("\033[92m", "\033[35m", "\033[33m", "\033[91m")

	gencfg: GenerationConfig
	decoder: embedding_decoder.EmbeddingDecoder
	vocab_targets_set: set
	vocab_targets: Optional[torch.Tensor]
	guide_targets_set: set
	guide_targets: Optional[torch.Tensor]
	class_lists: Optional[Sequence[Sequence[str]]] = None
	precompute: Optional[Any] = None
	target: Optional[torch.Tensor] = None
	target_padding: Optional[torch.Tensor] = None
	target_score: Optional[list] = None
	num_samples: int = 0
	target_str: Optional[list] = None
	invalid: Optional[torch.Tensor] = None
	valid_vocab: Optional[torch.Tensor] = None
	valid_guide: Optional[torch.Tensor] = None
	correct: Optional[torch.Tensor] = None
	result: Optional[torch.Tensor] = None
	topk_counts: torch.Tensor = dataclasses.field(init=False)
	topk_invalid: Optional[torch.Tensor] = None
	topk_valid: Optional[torch.Tensor] = None
	topk_vocab: Optional[torch.Tensor] = None
	topk_guide: Optional[torch.Tensor] = None
	topk: Optional[torch.Tensor] = None

	def __post_init__(self):
		self.topk_counts = torch.zeros((self.gencfg.topk, 4), dtype=torch.int64)
		if self.gencfg.vocab_prior and self.vocab_targets is None:
			raise ValueError("Generation config specifies to use vocab priors but no vocab targets were provided")
		if self.gencfg.guided and self.guide_targets is None:
			raise ValueError("Generation config is guided but no guide targets were provided")
		if self.gencfg.method == "greedy":
			if self.gencfg.topk != 1:
				raise ValueError(f"Top-k must be 1 for greedy generation: {self.gencfg.topk}")
			if self.gencfg.vocab_prior:
				raise ValueError("Greedy generation does not support vocab priors")
		elif self.gencfg.method == "all" and not self.gencfg.guided:
			raise ValueError(f"The '{self.gencfg.method}' generation method must always be guided")

	def clear(self, clear_precompute: bool = False):
		if clear_precompute:
			self.precompute = None
		self.target = self.target_padding = self.target_score = self.target_str = None
		self.invalid = self.valid_vocab = self.valid_guide = self.correct = self.result = None
		self.topk_invalid = self.topk_valid = self.topk_vocab = self.topk_guide = self.topk = None
		self.num_samples = 0
		self.topk_counts = torch.zeros((self.gencfg.topk, 4), dtype=torch.int64)

	def process(self, embeds: torch.Tensor, *, class_indices: Optional[Sequence[int]] = None, precompute: bool = True, precompute_cache=None):
		target, target_padding, target_score = self.generate(embeds=embeds, precompute=precompute, precompute_cache=precompute_cache)
		self.update(target=target, target_padding=target_padding, target_score=target_score, class_indices=class_indices)

	def generate(self, embeds: torch.Tensor, *, precompute: bool = True, precompute_cache=None):
		g = self.gencfg
		if g.method == "greedy":
			target, pad, _, _, _, score = self.decoder.generate(embed=embeds, collect_logits=False, calc_loss=True, temperature=g.temperature, length_alpha=g.length_alpha,
			                                                    sample_weight=None, guide_targets=self.guide_targets if g.guided else None, guide_renorm=g.guide_renorm)
			return target.unsqueeze(1), pad.unsqueeze(1), score.unsqueeze(1)
		if g.method == "beam":
			return self.decoder.generate_beam(embed=embeds, topk=g.topk, temperature=g.temperature, length_alpha=g.length_alpha,
			                                  vocab_targets=self.vocab_targets if g.vocab_prior else None, vocab_per_token=g.vocab_per_token, vocab_scaler=g.vocab_scaler,
			                                  guide_targets=self.guide_targets if g.guided else None, guide_renorm=g.guide_renorm)
		if g.method == "all":
			if precompute and self.precompute is None:
				self.precompute = self.decoder.precompute_generate_all(length_alpha=g.length_alpha, vocab_targets=self.vocab_targets if g.vocab_prior else None,
				                                                       vocab_per_token=g.vocab_per_token, vocab_scaler=g.vocab_scaler, guide_targets=self.guide_targets,
				                                                       guide_renorm=g.guide_renorm)
			return self.decoder.generate_all(embed=embeds, topk=g.topk, temperature=g.temperature, length_alpha=g.length_alpha,
			                                 vocab_targets=self.vocab_targets if g.vocab_prior else None, vocab_per_token=g.vocab_per_token, vocab_scaler=g.vocab_scaler,
			                                 guide_targets=self.guide_targets, guide_renorm=g.guide_renorm, precompute=self.precompute)
		raise ValueError(f"Unsupported generation method: {g.method}")

	def generate_many(self, embeds_list: Sequence[torch.Tensor], *, precompute: bool = True) -> list:
		"""generate() for several independent batches: greedy and beam search decode them concurrently (one stream + decode session per batch,
		embedding_decoder.generate_many / generate_beam_many: bit-identical to one call per batch); the teacher-forced 'all' method runs them one after the other."""
		g = self.gencfg
		same = len({tuple(e.shape) for e in embeds_list}) == 1
		if len(embeds_list) > 1 and same and g.method == "greedy":
			outs = self.decoder.generate_many(embeds_list, False, True, g.temperature, g.length_alpha, None, self.guide_targets if g.guided else None, g.guide_renorm)
			return [(o[0].unsqueeze(1), o[1].unsqueeze(1), o[5].unsqueeze(1)) for o in outs]
		if len(embeds_list) > 1 and same and g.method == "beam":
			return self.decoder.generate_beam_many(embeds_list, g.topk, g.temperature, g.length_alpha, self.vocab_targets if g.vocab_prior else None, g.vocab_per_token, g.vocab_scaler,
			                                       self.guide_targets if g.guided else None, g.guide_renorm)
		return [self.generate(e, precompute=precompute) for e in embeds_list]

	def update(self, target: torch.Tensor, target_padding: torch.Tensor, target_score: torch.Tensor, *, class_indices: Optional[Sequence[int]] = None):
		self.target = target.cpu()  # the one device->host transfer per batch
		self.target_padding = target_padding.cpu()
		self.target_score = target_score.tolist()
		self.num_samples += self.target.shape[0]
		self.target_str = self.decoder.embedder.detokenize_target(self.target)
		as_bool = lambda pred_set: torch.tensor([[p in pred_set for p in preds] for preds in self.target_str], dtype=torch.bool).reshape(self.target.shape[:-1])
		self.valid_vocab, self.valid_guide = as_bool(self.vocab_targets_set or ()), as_bool(self.guide_targets_set or ())
		if class_indices is not None and self.class_lists is not None:
			self.correct = torch.tensor([[p in self.class_lists[c] for p in preds] for c, preds in zip(class_indices, self.target_str)], dtype=torch.bool)
		else:
			self.correct = torch.zeros(self.target.shape[:-1], dtype=torch.bool)
		self.invalid = ~(self.valid_vocab | self.valid_guide | self.correct)
		stacked = torch.stack((self.correct, self.valid_guide, self.valid_vocab, torch.ones_like(self.invalid)), dim=2).cummax(dim=2)[0]
		self.result = torch.max(stacked.to(torch.uint8), dim=2)[1]  # first category that holds
		stacked[:, :, -1] = self.invalid
		self.topk_counts.add_(stacked.cummax(dim=1)[0].sum(dim=0))
		counts = self.topk_counts.to(torch.float32)
		self.topk_valid = (self.num_samples - counts[:, 3]) / self.num_samples
		ratios = counts / self.num_samples
		self.topk_invalid, self.topk_vocab, self.topk_guide, self.topk = ratios[:, 3], ratios[:, 2], ratios[:, 1], ratios[:, 0]


# ------------------------------------------------------------------------------------------------------------------------------
# loaders (reference infer.py:651-778)
# ------------------------------------------------------------------------------------------------------------------------------

def load_device(device: Union[torch.device, str, int]) -> tuple[torch.device, bool, bool]:
	device = torch.device(device)
	if device.type == "cuda" and not torch.cuda.is_available():
		raise RuntimeError("No MI355X device is available: this build has no CPU path for the decoder (the reference falls back to CPU here, infer.py:653-655)")
	device = torch.empty((), device=device).device
	return device, device.type == "cpu", device.type == "cuda"


def load_decoder_amp(enabled: bool, bf16: bool, determ: bool, device: torch.device):
	"""The HIP decoder always computes its GEMMs in bf16 with fp32 accumulation; there is no autocast context to enter."""
	return contextlib.nullcontext(), torch.bfloat16


def load_target_config(checkpoint: dict[str, Any], embedder: embedders.Embedder) -> embedders.TargetConfig:
	target_config = utils.dataclass_from_dict(embedders.TargetConfig, checkpoint["target_config"])
	embedder.configure_target(target_config=target_config, target_vocab=checkpoint["target_nouns"][checkpoint["num_invalid_target_nouns"]:])
	return target_config


def load_guide_targets(guide_targets: tuple, embedder: embedders.Embedder, device: torch.device, device_is_cpu: bool) -> torch.Tensor:
	assert isinstance(guide_targets, tuple) and guide_targets and all(isinstance(t, str) for t in guide_targets)
	if len(set(guide_targets)) != len(guide_targets):
		raise ValueError("Guide target nouns contain duplicates")
	tc = embedder.target_config
	ids = torch.full((len(guide_targets), tc.token_length), tc.pad_token_id, dtype=tc.token_dtype)
	bs = embedder.tokenizer_batch_size
	for i in range(0, len(guide_targets), bs):
		t = embedder.tokenize_target(guide_targets[i:i + bs])[0]
		if t.shape[1] > ids.shape[1]:
			raise ValueError("Some guide target noun(s) have tokenizations that are longer than supported by the model target configuration")
		ids[i:i + bs, :t.shape[1]] = t
	ids = ids[torch.all(ids >= 0, dim=1)]
	return ids if device_is_cpu else ids.to(device)


MODEL_KWARGS = ("vocab_quant", "num_end_loss", "label_smoothing", "hidden_dim", "feedfwd_scale", "mlp_hidden_layer", "mlp_hidden_bias", "mlp_hidden_norm", "mlp_hidden_activation",
                "input_dropout", "num_layers", "num_heads", "layer_dropout", "layer_activation", "layer_norm_first", "layer_bias", "logits_bias", "init_bias_zero",
                "init_mlp_mode", "init_mlp_unit_norm", "init_tfrm_mode", "init_tfrm_unit_norm", "init_tfrm_unit_postnorm", "init_tfrm_proj_layers", "init_zero_norm",
                "init_rezero_mode")
PREFIXED_KWARGS = ("mlp_seq_len", "weight_tying", "strictly_causal", "enable_nested")


def load_decoder_model(cfg: Any, embedder: embedders.Embedder, data_config: embedding_dataset.DataConfig, checkpoint: Optional[dict[str, Any]]) -> embedding_decoder.EmbeddingDecoder:
	model_class: Type[embedding_decoder.EmbeddingDecoder] = getattr(embedding_decoder, cfg.model)
	assert embedder.target_config is not None
	kwargs = dict(embedder=embedder, data_config=data_config, **{k: getattr(cfg, k) for k in MODEL_KWARGS})
	if model_class is embedding_decoder.PrefixedIterDecoder:
		kwargs.update({k: getattr(cfg, k) for k in PREFIXED_KWARGS})
	elif model_class is embedding_decoder.DudDecoder:  # the zero-parameter baseline (reference :759-762)
		kwargs.update(mlp_seq_len=cfg.mlp_seq_len)
	else:
		raise ValueError(f"Unrecognised model class: {model_class.__qualname__}")
	model = model_class(**kwargs)
	if checkpoint is not None:
		model.load_state_dict(checkpoint["model_state_dict"], strict=True)
	return model


# ------------------------------------------------------------------------------------------------------------------------------
# NOVICModel (reference infer.py:46-350)
# ------------------------------------------------------------------------------------------------------------------------------

def split_decode_groups(sizes: Sequence[int], limit: int):
	"""[(row range, [batch sizes])]: consecutive caller batches of one tower launch, cut into decode calls of at most `limit` rows (a batch is never split; one that is
	larger than the limit is a call of its own)."""
	out, start, at, part = [], 0, 0, []
	for n in sizes:
		if part and at - start + n > limit:
			out.append(((start, at), part))
			start, part = at, []
		part.append(n)
		at += n
	if part:
		out.append(((start, at), part))
	return out


class NOVICModel:

	def __init__(self, checkpoint: str, *, gencfg: str = "beam_k10_vnone_gp_t1_a0", guide_targets: Union[Iterable[str], str, None] = None, torch_compile: bool = False,
	             batch_size: int = 128, device: Union[torch.device, str, int] = "cuda", cfg_flat_override: Optional[dict[str, Any]] = None,
	             embedder_override: Optional[dict[str, Any]] = None, embedder: Optional[embedders.Embedder] = None):
		self.checkpoint = os.path.abspath(checkpoint)
		self.checkpoint_tail = os.path.join(os.path.basename(os.path.dirname(self.checkpoint)), os.path.basename(self.checkpoint))
		lazy = torch.load(self.checkpoint, map_location="cpu", mmap=True, weights_only=False)
		cfg_flat = dict(lazy["cfg_flat"])
		if cfg_flat_override is not None:
			cfg_flat.update(cfg_flat_override)
		self.cfg = utils.AttrDict.from_dict(utils.unflatten_dict(cfg_flat))
		del lazy
		self.gentask: Optional[GenerationTask] = None
		self.guide_targets_tensor = self.guide_targets_str_set = self.vocab_targets_tensor = self.model_targets_set = self.model_targets = None
		self.decoder: Optional[embedding_decoder.EmbeddingDecoder] = None
		self.set_gencfg(gencfg, update_task=False)
		self.set_guide_targets(guide_targets, update_task=False)
		self.torch_compile = torch_compile  # accepted for signature compatibility; there is nothing to compile
		self.batch_size = batch_size
		self.device, self.device_is_cpu, self.device_is_cuda = load_device(device)
		if embedder is not None:
			self.embedder = embedder
		else:
			kw = dict(spec=self.cfg.embedder_spec, amp=self.cfg.embedder_amp, amp_bf16=self.cfg.embedder_amp_bf16, tokenizer_batch_size=batch_size, inference_batch_size=batch_size,
			          image_batch_size=batch_size, load_model=False, compile_model=self.cfg.get("embedder_compile", False), use_optimum=self.cfg.get("embedder_optimum", False),
			          device=self.device, check=False)
			if embedder_override is not None:
				kw.update(embedder_override)
			self.embedder = embedders.Embedder.create(**kw)
		self.amp_context, self.amp_dtype = load_decoder_amp(enabled=self.cfg.get("amp", True), bf16=self.cfg.get("amp_bf16", True), determ=False, device=self.device)
		self.data_config = embedding_dataset.DataConfig.create(dict(use_weights=False, unit_weights=True, multi_target=False, multi_first=False, full_targets=True,
		                                                           fixed_multi_length=True, multi_length=1), use_targets=True)
		self.__stack = contextlib.ExitStack()

	def set_gencfg(self, gencfg: str, update_task: bool = True):
		self.gencfg = GenerationConfig.from_name(gencfg)
		if update_task:
			self.update_gentask()

	def set_guide_targets(self, guide_targets: Union[Iterable[str], str, None] = None, update_task: bool = True):
		if guide_targets is None:
			self.guide_targets = None
		elif isinstance(guide_targets, str):
			with open(guide_targets, "r") as f:
				self.guide_targets = tuple(s for line in f if (s := line.strip()))
		else:
			self.guide_targets = tuple(guide_targets)
		self.update_guide_targets()
		if update_task:
			self.update_gentask()

	def set_batch_size(self, batch_size: int):
		self.batch_size = batch_size
		self.embedder.tokenizer_batch_size = self.embedder.inference_batch_size = self.embedder.image_batch_size = batch_size

	@contextlib.contextmanager
	def decoder_model(self, release=True):
		if self.is_decoder_loaded():
			yield
		else:
			try:
				self.load_decoder()
				yield
			finally:
				self.unload_decoder()

	def load_decoder(self) -> bool:
		if self.decoder is not None:
			return False
		ckpt = torch.load(self.checkpoint, map_location="cpu", weights_only=False)
		load_target_config(ckpt, self.embedder)
		self.model_targets = self.embedder.target_vocab
		self.model_targets_set = set(self.model_targets)
		self.vocab_targets_tensor = self.embedder.tokenize_target(self.model_targets)[0].to(self.device)
		self.update_guide_targets()
		with torch.inference_mode():
			self.decoder = load_decoder_model(self.cfg, self.embedder, self.data_config, ckpt)
			self.decoder.to(self.device)
			self.decoder.eval()
		del ckpt
		gc.collect()
		self.update_gentask()
		return True

	def update_guide_targets(self):
		if self.model_targets is None:
			self.guide_targets_str_set = self.guide_targets_tensor = None
		else:
			names = self.guide_targets if self.guide_targets is not None else self.model_targets
			self.guide_targets_str_set = set(names)
			self.guide_targets_tensor = load_guide_targets(tuple(names), self.embedder, self.device, self.device_is_cpu)

	def update_gentask(self):
		self.gentask = None if self.decoder is None else GenerationTask(gencfg=self.gencfg, decoder=self.decoder, vocab_targets_set=self.model_targets_set,
		                                                               vocab_targets=self.vocab_targets_tensor, guide_targets_set=self.guide_targets_str_set,
		                                                               guide_targets=self.guide_targets_tensor, class_lists=None)

	def unload_decoder(self) -> bool:
		if self.decoder is None:
			return False
		self.gentask = self.guide_targets_tensor = self.guide_targets_str_set = self.vocab_targets_tensor = self.model_targets_set = self.model_targets = self.decoder = None
		return True

	def is_decoder_loaded(self) -> bool:
		return self.decoder is not None

	@contextlib.contextmanager
	def inference_mode(self):
		with torch.inference_mode():
			yield

	@classmethod
	def load_image(cls, image_path: str):
		import PIL.Image
		image = PIL.Image.open(image_path)
		image.load()
		return image if image.mode == "RGB" else image.convert("RGB")

	@classmethod
	def load_images(cls, image_paths: Iterable[str], *, image_dir: Optional[str] = None):
		return [cls.load_image(os.path.join(image_dir or "", p)) for p in image_paths]

	def load_image_batches(self, image_paths: Iterable[str], *, image_dir: Optional[str] = None, batch_size: Optional[int] = None):
		bs = batch_size or self.batch_size
		it = iter(image_paths)
		out = []
		while chunk := tuple(itertools.islice(it, bs)):
			out.append([self.load_image(os.path.join(image_dir or "", p)) for p in chunk])
		return out

	# PIL images -> uint8 pixel batches, normalised on the device by the tower's first kernel (Embedder.get_image_transform(uint8=True): same fp32 arithmetic, bit-identical
	# embeddings, a quarter of the host -> device bytes).  Off: the reference's interface moves normalised fp32 images (embedders.py:755-764) and stays the default.
	uint8_images = False

	def get_image_transform(self, uint8: Optional[bool] = None) -> Callable:
		if self.uint8_images if uint8 is None else uint8:
			return self.embedder.get_image_transform(uint8=True)
		return self.embedder.get_image_transform()

	def transform_images(self, images) -> torch.Tensor:
		import PIL.Image
		if isinstance(images, PIL.Image.Image):
			images = (images,)
		tf = self.get_image_transform()
		return torch.stack([tf(im) for im in images], dim=0)

	def __enter__(self) -> "NOVICModel":
		with self.__stack as stack:
			stack.enter_context(self.embedder.inference_model(release=True))
			stack.enter_context(self.decoder_model(release=False))
			self.__stack = stack.pop_all()
		return self

	def __exit__(self, exc_type, exc_val, exc_tb) -> bool:
		return self.__stack.__exit__(exc_type, exc_val, exc_tb)

	def embed_images(self, images) -> torch.Tensor:
		if not isinstance(images, torch.Tensor):
			images = self.transform_images(images)
		with self.embedder.inference_mode():
			return self.embedder.inference_image(images=images)

	def classify_embeds(self, embeds: torch.Tensor) -> NOVICOutput:
		with self.inference_mode():
			self.gentask.process(embeds=embeds)
		t = self.gentask
		return NOVICOutput(embeds=embeds.cpu(), preds=tuple(tuple(" ".join(s.split()) for s in row) for row in t.target_str),
		                   logprobs=tuple(tuple(row) for row in t.target_score), probs=tuple(tuple(math.exp(s) for s in row) for row in t.target_score),
		                   types=tuple(tuple(PredictionType(r) for r in row) for row in t.result.tolist()))

	# Rows one decode call takes when consecutive caller batches share a tower launch (classify_image_batches).  A decode step at 256 rows is launch-latency-bound (~200 us for
	# ~33 launches whatever the rows: greedy 2.2 ms per 256 rows, 3.2 ms per 1 024), so the caller batches of one tower launch are decoded together.  A sample's result does
	# not depend on the rows it shares a call with -- the rows of the decode GEMMs and the per-sequence attention / step kernels never mix, and since round 5 the two
	# row-count regimes of the layer step (LayerNorm as a GEMM prologue up to 512 rows, as its own launch beyond) run the same IEEE operation sequence
	# (csrc/common.hpp `unfused`; tools/decode_rows_identity.py) -- so the labels, scores and paddings are those of one call per batch, bit for bit.
	decode_rows = 1024
	# Decode calls in flight at the same time (classify_image_batches): beside a running tower a decode call is latency-bound three to four times over (a 1 024-row greedy
	# call 3.0 ms alone, 12.2 ms beside ViT-B/32 -- longer than the tower's own 10.9 ms per 1 024 images, so the pipeline waited for the decoder); two calls on lanes of their
	# own (GenerationTask.generate_many: bit-identical to one call each) hide each other's latency: 87.4 k -> 92.2 k labels/s on long runs (tools/e2e_timeline.py,
	# tools/e2e_lanes_sweep.sh; three lanes: 79 k).  For batches that are ALREADY ON THE DEVICE: with batches staged from the host the second lane loses what it wins and more
	# (pinned uint8 batches 83.7 k labels/s one call at a time, 76-80 k on two lanes; pinned fp32 81.5 k -> 71.4 k) -- a fifth stream (the H2D copies') beside main, tower and two
	# lanes, on a runtime whose queue-to-stream binding follows the process's history (the same pipeline inside bench.py's process, which had used other streams before, ran
	# 88.9 k; with GPU_MAX_HW_QUEUES=4 the stand-alone tool ran 90.3 k) -- so host batches keep one decode call at a time.
	decode_lanes = 2

	def classify_embeds_many(self, embeds_list: Sequence[torch.Tensor]) -> list:
		"""classify_embeds for several independent batches, decoded concurrently where the generation method allows (GenerationTask.generate_many): the outputs of one call each."""
		with self.inference_mode():
			gens = self.gentask.generate_many(list(embeds_list))
		outs = []
		t = self.gentask
		for embeds, (target, pad, score) in zip(embeds_list, gens):
			t.update(target=target, target_padding=pad, target_score=score)
			outs.append(NOVICOutput(embeds=embeds.cpu(), preds=tuple(tuple(" ".join(s.split()) for s in row) for row in t.target_str),
			                        logprobs=tuple(tuple(row) for row in t.target_score), probs=tuple(tuple(math.exp(s) for s in row) for row in t.target_score),
			                        types=tuple(tuple(PredictionType(r) for r in row) for row in t.result.tolist())))
		return outs

	def classify_image_batches(self, batches, persistent_cus: Optional[int] = None, coalesce: Optional[int] = None, decode_rows: Optional[int] = None,
	                           latency: bool = False, decode_lanes: Optional[int] = None) -> Iterator[NOVICOutput]:
		"""`classify_images` over consecutive batches (tensors or lists of PIL images) with the image tower of the next batch(es) running beside the decoding of the current one
		(`Embedder.inference_image_batches`: consecutive batches of one shape share a tower launch, `coalesce`, and up to `decode_rows` rows of them a decode call); yields one
		NOVICOutput per CALLER batch, the same predictions as one call per batch -- a coalesced tower launch is only chosen where its embeddings are the single-batch ones bit
		for bit (no K-split tail in either launch: `Embedder.inference_image_batches`), and a decode result does not depend on the rows it shares a call with.
		Throughput mode by default: up to sixteen batches are taken from `batches` before the first output appears (two tower launches of four in flight, two staged) and, for
		batches that are already on the device, `decode_lanes` = 2 launches' embeddings are decoded at the same time -- each on a lane of its own, the results those of one call
		each; batches staged from the host are decoded one call at a time (see `decode_lanes`).  latency = True: one batch per
		tower launch and per decode call, one call at a time, two batches of look-ahead -- for a live or slow producer."""
		tensors = (b if isinstance(b, torch.Tensor) else self.transform_images(b) for b in batches)
		limit = int(self.decode_rows if decode_rows is None else decode_rows)
		lanes = max(1, int(self.decode_lanes if decode_lanes is None else decode_lanes))
		if latency:
			limit = lanes = 1  # (a batch is never split: every caller batch becomes a decode call of its own)

		def hand_out(out, part):
			if len(part) == 1:
				yield out
				return
			at = 0
			for n in part:  # per-sample tuples: a caller batch's share is a slice
				yield NOVICOutput(embeds=out.embeds[at:at + n], preds=out.preds[at:at + n], logprobs=out.logprobs[at:at + n], probs=out.probs[at:at + n], types=out.types[at:at + n])
				at += n

		on_host = [False]

		def watch(ts):  # (where the caller's batches live decides the lanes: see decode_lanes)
			for t in ts:
				if t.device.type == "cpu":
					on_host[0] = True
				yield t

		held = []  # decode groups waiting for their lane mates: (embeddings, caller batch sizes)
		for embeds, sizes in self.embedder.inference_image_batches(watch(tensors), persistent_cus=persistent_cus, coalesce=coalesce, grouped=True, latency=latency, decode_lanes=lanes):
			for rows, part in split_decode_groups(sizes, limit):
				held.append((embeds[rows[0]:rows[1]], part))
				if len(held) < (1 if on_host[0] else lanes):
					continue
				outs = self.classify_embeds_many([h[0] for h in held]) if len(held) > 1 else [self.classify_embeds(held[0][0])]
				for out, (_, p) in zip(outs, held):
					yield from hand_out(out, p)
				held = []
		if held:
			outs = self.classify_embeds_many([h[0] for h in held]) if len(held) > 1 else [self.classify_embeds(held[0][0])]
			for out, (_, p) in zip(outs, held):
				yield from hand_out(out, p)

	def classify_image(self, image) -> NOVICOutput:
		return self.classify_images(image)

	def classify_images(self, images) -> NOVICOutput:
		return self.classify_embeds(self.embed_images(images))

	def __call__(self, images) -> NOVICOutput:
		return self.classify_images(images)


def main():
	import argparse
	parser = argparse.ArgumentParser(description="Inference a NOVIC model checkpoint on given image(s).")
	parser.add_argument("--checkpoint", type=str, required=True, metavar="CKPT")
	parser.add_argument("--image_dir", type=str, default=None, metavar="DIR")
	parser.add_argument("--images", type=str, nargs="+", required=True, metavar="PATH")
	parser.add_argument("--gencfg", type=str, default="beam_k10_vnone_gp_t1_a0", metavar="GENCFG")
	parser.add_argument("--guide_targets", type=str, nargs="+", default=None, metavar="NOUN")
	parser.add_argument("--guide_targets_file", type=str, default=None, metavar="PATH")
	parser.add_argument("--torch_compile", action="store_true")
	parser.add_argument("--batch_size", type=int, default=128, metavar="NUM")
	parser.add_argument("--device", type=str, default="cuda", metavar="DEV")
	parser.add_argument("--no_tf32", dest="tf32", action="store_false")
	args = parser.parse_args()
	if args.guide_targets is not None and args.guide_targets_file is not None:
		parser.error("Cannot specify both --guide_targets and --guide_targets_file")
	model = NOVICModel(checkpoint=args.checkpoint, gencfg=args.gencfg, guide_targets=args.guide_targets_file or args.guide_targets, torch_compile=args.torch_compile,
	                   batch_size=args.batch_size, device=args.device)
	batches = model.load_image_batches(args.images, image_dir=args.image_dir)
	with model:
		lines = []
		for out in model.classify_image_batches(batches):
			lines.extend(" / ".join(f"{GenerationTask.COLOR_MAP[t.value]}{p}\033[0m = {pr * 100:.3g}%" for p, pr, t in itertools.islice(zip(ps, prs, ts), 3))
			             for ps, prs, ts in zip(out.preds, out.probs, out.types))
		for path, line in zip(args.images, lines):
			print(f"{path} --> {line}")


if __name__ == "__main__":
	main()

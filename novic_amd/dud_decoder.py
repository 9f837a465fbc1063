"""`DudDecoder`: the reference's zero-parameter baseline "decoder" (reference embedding_decoder.py:454-610; dispatched by `infer.load_decoder_model`, infer.py:759).

It has nothing to compute on the GPU and nothing to train: `forward` CHEATS from the targets it is handed (one-hot "logits" of a prediction derived from the targets
themselves), `generate*` always answer with the tokens of the word 'unknown'.  The reference keeps it as the floor of its evaluation tables, and a checkpoint whose
`cfg.model` names it must load -- that is why it exists here.  HOST bookkeeping only: a handful of torch index operations on whatever device the tensors live on, no entry
point of the C ABI, not part of the hot path (SURVEY.md section 8 lists no row for it; the round-5 review asked for it as a loader completeness item).

The multi-target rule of `forward`, restated (reference :498-527): targets of one embedding are visited in order m = 0 .. M-2; among the targets r >= m whose tokens BEFORE
position c equal target m's (and whose position c is not padding) the token at c with the largest total weight -- count without weights -- wins, the lowest id on a tie,
and every member of that group predicts the winner at c.  Votes are cast with the predictions as they stand (an earlier m may already have rewritten them), membership is
decided on the original targets.  The reference tallies the votes in a (V + 1)-bin histogram per position; here the tally is taken over the <= M members directly (the
winner is always a member's token), adding the weights in the same order r = m, m+1, ... so that ties fall the same way.
"""
from __future__ import annotations

import math
from typing import Any, Optional

import torch

from .embedding_decoder import EmbeddingDecoder, ParamCount


class DudDecoder(EmbeddingDecoder):

	@classmethod
	def get_target_config_kwargs(cls, **target_kwargs) -> dict[str, Any]:
		return target_kwargs  # (no demands on the target configuration: reference :456-458)

	@classmethod
	def get_data_config_kwargs(cls, **data_kwargs) -> dict[str, Any]:
		return data_kwargs

	def __init__(self, **kwargs):
		super().__init__(**kwargs)
		ids, pad = self.embedder.tokenize_target("unknown")
		if bool((ids < 0).any()):  # 'unknown' is not spelt by the compact vocabulary: fall back to the empty noun (reference :467-468)
			ids, pad = self.embedder.tokenize_target("")
		assert ids.shape == pad.shape and ids.shape[0] == 1 and ids.shape[1] >= 1 and not bool(pad.any())
		self.dud_target, self.dud_target_padding = ids, pad

	def get_num_params(self):
		none = ParamCount(total=0, used=0, unused=0, trained=0, frozen=0)
		return none, {"Dud": none}

	# ---- forward: predictions read off the targets ----
	def _loss_padding(self, target: torch.Tensor, padding: Optional[torch.Tensor], weight: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
		"""The padding the outputs carry: zero-weighted targets are padding throughout; with num_end_loss = N > 1 the first N - 1 positions take position 0's flag and the
		rest is the given mask shifted right by N - 1 (reference :484-496)."""
		if weight is not None:
			dead = (weight == 0).unsqueeze(-1)
			padding = dead.expand_as(target) if padding is None else padding | dead
		if padding is None or self.num_end_loss <= 1:
			return padding
		C, lead = target.shape[-1], self.num_end_loss - 1
		keep = C - lead
		first = padding[..., :1]
		if keep <= 1:
			return first.expand_as(target)
		return torch.cat((first.expand(*target.shape[:-1], lead), padding[..., :keep]), dim=-1)

	@staticmethod
	def _vote(target: torch.Tensor, padding: Optional[torch.Tensor], weight: Optional[torch.Tensor]) -> torch.Tensor:
		"""target [B][M][C] (ids), padding [B][M][C] or None, weight [B][M] or None -> the predictions [B][M][C] of the rule in the module docstring."""
		B, M, C = target.shape
		pred = target.clone()
		big = torch.iinfo(target.dtype).max
		for m in range(M - 1):
			tail = target[:, m:, :]
			agree = (tail[:, :, :-1] == tail[:, :1, :-1]).to(torch.int8).cumprod(dim=-1).bool()  # [B][R][C-1]: every token before position c+1 equals target m's
			member = torch.cat((torch.ones_like(agree[:, :, :1]), agree), dim=-1)  # position 0 has an empty prefix
			if padding is not None:
				member = member & ~padding[:, m:, :]
			cur = pred[:, m:, :]  # (a view: rewritten below)
			R = cur.shape[1]
			tally = torch.zeros(cur.shape, dtype=torch.int64 if weight is None else weight.dtype, device=target.device)
			for r in range(R):  # the weight of voter r goes to everybody who holds voter r's token, voters in ascending order (the order a histogram would add them in)
				share = member[:, r:r + 1, :] & (cur == cur[:, r:r + 1, :])
				if weight is None:
					tally += share
				else:
					tally += share * weight[:, m + r, None, None]
			if weight is None:
				masked = torch.where(member, tally, torch.full_like(tally, -1))
			else:
				masked = torch.where(member, tally, torch.full_like(tally, -math.inf))
			best = masked.max(dim=1, keepdim=True)[0]
			leads = member & (masked == best)
			winner = torch.where(leads, cur, torch.full_like(cur, big)).min(dim=1, keepdim=True)[0]  # lowest token id among the leaders
			pred[:, m:, :] = torch.where(member, winner.expand_as(cur), cur)
		return pred

	def forward(self, embed, target, target_padding, target_weight, calc_loss: bool, calc_correct: bool, only_pred: bool, guide_targets):
		# (guide targets are ignored, as in the reference)
		if target is None:
			raise ValueError(f"{self.__class__.__name__} can only cheat, so it requires targets that it can cheat from")
		V = self.target_config.vocab_size
		padding = self._loss_padding(target, target_padding, target_weight)
		if target.ndim == 3:
			mf = bool(self.data_config.multi_first)
			swap = (lambda t: t.transpose(0, 1)) if mf else (lambda t: t)
			pred = swap(self._vote(swap(target), None if padding is None else swap(padding), None if target_weight is None else swap(target_weight)))
			pred = pred.contiguous()
		else:
			pred = target.clone()
		logits = torch.zeros(*pred.shape, V, dtype=embed.dtype, device=embed.device)
		logits.scatter_(-1, pred.unsqueeze(-1), 1.0)
		if only_pred:
			pred, logits, target = pred[..., -1:], logits[..., -1:, :], target[..., -1:]
			if padding is not None:
				padding = padding[..., -1:]
		one = (lambda: torch.ones((), dtype=embed.dtype, device=embed.device)) if calc_loss else (lambda: None)
		correct = None
		if calc_correct:
			correct = pred == target
			if padding is not None:
				correct = correct & ~padding
		return logits, padding, one(), one(), correct

	# ---- generation: always 'unknown' ----
	def _dud(self, device):
		return self.dud_target.to(device), self.dud_target_padding.to(device)

	def generate(self, embed, collect_logits: bool, calc_loss: bool, temperature: float, length_alpha: float, sample_weight, guide_targets, guide_renorm: bool):
		B, V = embed.shape[0], self.target_config.vocab_size
		ids, pad = self._dud(embed.device)
		C = ids.shape[1]
		target, padding = ids.expand(B, -1).contiguous(), pad.expand(B, -1).contiguous()
		logits = None
		if collect_logits or calc_loss:
			logits = torch.zeros(B, C, V, dtype=embed.dtype, device=embed.device)
			logits.scatter_(-1, target.unsqueeze(-1), 1.0)
		if not calc_loss:
			return target, padding, logits, None, None, None
		# one-hot logits (1 at the token, 0 elsewhere) in closed form: log-softmax(x / t) at the token = 1/t - log(e^(1/t) + V - 1); cross entropy with smoothing s per
		# token = (1 - s) (L - 1) + s (L - 1/V) with L = log(e + V - 1)  (reference :560-570 builds both from the B x C x V tensor)
		it = 1.0 / temperature
		logp = it - math.log(math.exp(it) + V - 1)
		score = torch.full((B,), C * logp * (math.pow(C, -length_alpha) if length_alpha != 0 else 1.0), dtype=embed.dtype, device=embed.device)
		L, s = math.log(math.e + V - 1), float(self.label_smoothing)
		loss_sum = torch.tensor(B * C * ((1 - s) * (L - 1) + s * (L - 1 / V)), dtype=embed.dtype, device=embed.device)
		loss_basis = torch.tensor(B * C, dtype=embed.dtype, device=embed.device)
		return target, padding, logits, loss_sum, loss_basis, score

	def _single_beam(self, embed, topk: int, width: int):
		"""[B][topk][width] ids / padding / scores with ONE valid beam -- the dud noun, score -1 -- and the rest padded out at -inf (reference :580-591, :599-610)."""
		B = embed.shape[0]
		ids, pad = self._dud(embed.device)
		C = ids.shape[1]
		target = torch.zeros(B, topk, width, dtype=ids.dtype, device=embed.device)
		padding = torch.ones(B, topk, width, dtype=pad.dtype, device=embed.device)
		score = torch.full((B, topk), -math.inf, dtype=embed.dtype, device=embed.device)
		target[:, 0, :C], padding[:, 0, :C], score[:, 0] = ids, pad, -1.0
		return target, padding, score

	def generate_beam(self, embed, topk: int, temperature: float, length_alpha: float, vocab_targets, vocab_per_token: bool, vocab_scaler: float, guide_targets, guide_renorm: bool):
		return self._single_beam(embed, topk, self.dud_target.shape[1])

	# (the batch-concurrent entries GenerationTask.generate_many uses: nothing to overlap here, one call per batch)
	def generate_many(self, embeds, *args):
		return [self.generate(e, *args) for e in embeds]

	def generate_beam_many(self, embeds, *args):
		return [self.generate_beam(e, *args) for e in embeds]

	def precompute_generate_all(self, length_alpha: float, vocab_targets, vocab_per_token: bool, vocab_scaler: float, guide_targets, guide_renorm: bool):
		return None

	def generate_all(self, embed, topk: int, temperature: float, length_alpha: float, vocab_targets, vocab_per_token: bool, vocab_scaler: float, guide_targets, guide_renorm: bool,
	                 precompute=None):
		return self._single_beam(embed, topk, guide_targets.shape[1])

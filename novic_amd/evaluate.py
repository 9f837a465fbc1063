"""Evaluation / inference callers around the decoder (reference train.py:170-240 GenerationTaskList, :1726-1868 eval_top1_single, :2337-2450
eval_cls_decoding_single, :2606-2724 infer_model), without the reference's hydra / wandb / tqdm plumbing: the same loops and the same
aggregations over the same decoder calls, returning the same tuples / dictionaries, so the numbers a reference run logs can be reproduced.
"""
from __future__ import annotations

import dataclasses
import json
import math
import time
from typing import Iterator, Optional, Sequence, Union

import torch

from . import embedding_decoder, infer


class GenerationTaskList:
	"""All generation configs of an evaluation against one decoder (reference train.py:170-240)."""

	def __init__(self, gencfgs: Sequence[infer.GenerationConfig], model: embedding_decoder.EmbeddingDecoder, vocab_targets_set: set, vocab_targets: Optional[torch.Tensor],
	             guide_targets_set: set, guide_targets: Optional[torch.Tensor], class_lists: Optional[Sequence[Sequence[str]]] = None):
		self.gencfgs, self.model = tuple(gencfgs), model
		self.tasks = tuple(infer.GenerationTask(gencfg=g, decoder=model, vocab_targets_set=vocab_targets_set, vocab_targets=vocab_targets, guide_targets_set=guide_targets_set,
		                                        guide_targets=guide_targets, class_lists=class_lists) for g in self.gencfgs)

	def __len__(self) -> int:
		return len(self.tasks)

	def __getitem__(self, index: int) -> infer.GenerationTask:
		return self.tasks[index]

	def __iter__(self) -> Iterator[infer.GenerationTask]:
		return iter(self.tasks)

	def iter_generate(self, embeds: torch.Tensor, targets: Union[torch.Tensor, Sequence[int], None] = None):
		"""Task i+1 is generated on the GPU while task i's output is detokenised / scored on the host (the reference's prev_task pipelining, :203-231)."""
		if isinstance(targets, torch.Tensor):
			targets = targets.tolist()
		prev_task = prev_out = None
		for i, task in enumerate(self.tasks, 1):
			yield i, task
			out = task.generate(embeds=embeds)
			if prev_task is not None:
				prev_task.update(*prev_out, class_indices=targets)
			prev_task, prev_out = task, out
		if prev_task is not None:
			prev_task.update(*prev_out, class_indices=targets)

	def generate(self, embeds: torch.Tensor, targets=None):
		for _ in self.iter_generate(embeds, targets):
			pass

	def generate_many(self, embeds_list: Sequence[torch.Tensor], targets_list: Optional[Sequence] = None, on_batch=None):
		"""Several independent batches through every generation config, the batches of a config decoded CONCURRENTLY (GenerationTask.generate_many); the host-side
		detokenising / scoring of config i's outputs runs while config i + 1 decodes, as in iter_generate.  The tasks' counters see the batches in order, so the
		statistics equal those of generate() called once per batch.  on_batch(batch_index, outputs_of_that_batch) -- outputs_of_that_batch[task] = (target, padding,
		score) -- is called once per batch, in order, after EVERY task has been updated with every batch of this call.  The tasks' own per-batch state (target_str, ...)
		is the LAST batch's at that point: a caller that reads it per batch passes one batch per call (eval_cls_decoding(lanes=1)) or reads the outputs it is handed.
		Returns the list [task][batch] of (target, padding, score)."""
		targets_list = [None] * len(embeds_list) if targets_list is None else [t.tolist() if isinstance(t, torch.Tensor) else t for t in targets_list]
		outputs, prev = [], None
		for task in self.tasks:
			outs = task.generate_many(embeds_list)
			outputs.append(outs)
			if prev is not None:
				for out, tg in zip(prev[1], targets_list):
					prev[0].update(*out, class_indices=tg)
			prev = (task, outs)
		if prev is not None:
			for out, tg in zip(prev[1], targets_list):
				prev[0].update(*out, class_indices=tg)
		if on_batch is not None:
			for b in range(len(embeds_list)):
				on_batch(b, [outs[b] for outs in outputs])
		return outputs


def eval_top1(model: embedding_decoder.EmbeddingDecoder, loader, data_config, token_length: int, guide_token_ids: Optional[torch.Tensor] = None):
	"""Teacher-forced top-1 evaluation over one epoch of an embedding loader yielding (embed, target, mask, weight) device batches (reference :1755-1868).
	Returns (loss, noun top-1, token top-1, token top-1 per sequence position, tokens predicted, valid targets, samples, batches, seconds)."""
	num_batches = num_samples = num_valid_targets = samples_correct = 0
	correct_seq = torch.zeros(token_length, dtype=torch.int64)
	tokens_seq = torch.zeros(token_length, dtype=torch.int64)
	loss_sum_sum = loss_basis_sum = 0.0
	start = time.perf_counter()
	with torch.inference_mode():
		for embed, target, mask, weight in loader:
			_, padding, loss_sum, loss_basis, correct = model(embed=embed, target=target, target_padding=mask, target_weight=weight, calc_loss=True, calc_correct=True, only_pred=False,
			                                                  guide_targets=guide_token_ids)
			multi_dim = None if not data_config.multi_target else 0 if data_config.multi_first else 1
			multi_dims = target.shape[:-1]
			lead = tuple(range(correct.ndim - 1))
			batch_correct_seq = correct.sum(dim=lead)
			if padding is not None:
				valid_targets = ~padding.all(dim=-1)
				padding_seq = padding.sum(dim=lead)
				correct = correct | padding
			sample_correct = correct.all(dim=-1)
			if padding is not None:
				sample_correct = sample_correct & valid_targets
			if multi_dim is not None:
				sample_correct = sample_correct.any(dim=multi_dim)
			# one transfer per batch instead of the reference's five .item() round trips
			scal = torch.stack((loss_sum.float().reshape(()), torch.as_tensor(loss_basis, device=embed.device).float().reshape(()), sample_correct.sum().float(),
			                    (valid_targets.sum() if padding is not None else torch.zeros((), device=embed.device)).float())).tolist()
			num_batches += 1
			num_samples += embed.shape[0]
			loss_sum_sum += scal[0]
			loss_basis_sum += scal[1]
			samples_correct += int(scal[2])
			num_batch_targets = math.prod(multi_dims)
			num_valid_targets += num_batch_targets if padding is None else int(scal[3])
			C = target.shape[-1]
			correct_seq[:C] += batch_correct_seq.cpu()
			tokens_seq[:C] += num_batch_targets if padding is None else (num_batch_targets - padding_seq.cpu())
	elapsed = time.perf_counter() - start
	tokens_total = int(tokens_seq.sum())
	return (loss_sum_sum / loss_basis_sum, samples_correct / max(1, num_samples), int(correct_seq.sum()) / max(1, tokens_total), (correct_seq / tokens_seq).tolist(), tokens_total,
	        num_valid_targets, num_samples, num_batches, elapsed)


def eval_cls_decoding(task_list: GenerationTaskList, dataset_batches, device: torch.device, lanes: int = 3):
	"""dataset_batches: iterable of (embeds B x F, class indices, paths | None).  Returns per generation config (gencfg, top-k correct, top-k valid-guide, top-k
	valid-vocab, top-k invalid) ratio tensors (reference :2384-2450).  lanes: how many batches are decoded concurrently (1 = the reference's one-at-a-time loop; the
	statistics are the same either way: the batches are independent and every lane's outputs are bit-identical to its own call)."""
	for task in task_list:
		task.clear()
	with torch.inference_mode():
		group = []

		def flush():
			if len(group) == 1:
				task_list.generate(group[0][0], group[0][1])
			elif group:
				task_list.generate_many([e for e, _ in group], [t for _, t in group])
			group.clear()
		for embeds, targets, _paths in dataset_batches:
			if embeds.device != device:
				embeds = embeds.pin_memory().to(device, non_blocking=True) if embeds.device.type == "cpu" else embeds.to(device)
			if group and embeds.shape != group[0][0].shape:  # a ragged last batch decodes on its own
				flush()
			group.append((embeds, targets))
			if len(group) >= max(1, lanes):
				flush()
		flush()
	return tuple((task.gencfg, task.topk, task.topk_guide, task.topk_vocab, task.topk_invalid) for task in task_list)


def infer_predictions(task_list: GenerationTaskList, data) -> dict:
	"""data: iterable of (keys, embeds).  Returns {gencfg name: {key: ((noun, score, result), ...)}} (reference :2640-2656)."""
	predictions = {task: {} for task in task_list}
	with torch.inference_mode():
		for keys, embeds in data:
			task_list.generate(embeds=embeds)
			for task, preds in predictions.items():
				for key, nouns, scores, results in zip(keys, task.target_str, task.target_score, task.result.tolist()):
					preds[key] = tuple((" ".join(n.split()), s, r) for n, s, r in zip(nouns, scores, results))
	return {task.gencfg.name: preds for task, preds in predictions.items()}


def write_pred_json(path: str, task_list: GenerationTaskList, predictions: dict, *, model_path: str, guide_targets: Sequence[str], vocab_targets: Sequence[str], model_cfg=None,
                    infer_cfg=None) -> str:
	"""The predictions JSON of the reference's infer action (version 1 layout, :2693-2720)."""
	samples = tuple(next(iter(predictions.values())).keys()) if predictions else ()
	doc = dict(version=1, model=model_path, model_path=model_path, model_cfg=model_cfg, infer_cfg=infer_cfg, guide_targets=sorted(set(guide_targets)), vocab_targets=sorted(set(vocab_targets)),
	           samples=samples, predictions={})
	for task in task_list:
		preds = predictions[task.gencfg.name]
		doc["predictions"][task.gencfg.name] = dict(
			gen_cfg=dataclasses.asdict(task.gencfg), valid_guide=(task.topk_guide * 100).tolist(), valid_vocab=(task.topk_vocab * 100).tolist(), valid=(task.topk_valid * 100).tolist(),
			invalid=(task.topk_invalid * 100).tolist(), pred=tuple(tuple(i[0] for i in t) for t in preds.values()), score=tuple(tuple(i[1] for i in t) for t in preds.values()),
			result=tuple(tuple(i[2] for i in t) for t in preds.values()))
	with open(path, "w") as f:
		json.dump(doc, f, indent=2)
	return path

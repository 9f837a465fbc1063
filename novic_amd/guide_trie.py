"""Token trie over a set of tokenised nouns (guide targets / vocabulary targets) for guided decoding.

The reference tracks, per beam, a boolean mask over ALL W nouns ("still consistent with what was generated", embedding_decoder.py:788,
:877-879, :969-975) and rebuilds a (V+1)-wide allowed-token mask from it at every step (:808-809, :916-917).  Equivalent and O(children)
instead of O(W): a beam carries one trie node; the tokens some still-consistent noun has at the next column are exactly the node's children.
Edge payloads also carry the vocabulary prior of :924-936: log(count(child) / count(node)) per target, log(1 / #children) per token.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class TokenTrie:
	def __init__(self, targets: torch.Tensor, device: torch.device):
		"""targets: W x Cmax integer tensor, each row = content tokens, END (0), zero padding (tokenize_target output / load_guide_targets)."""
		rows = targets.detach().cpu().numpy()
		children: list[dict[int, int]] = [{}]   # node -> {token: child node | -1 for END}
		counts: list[dict[int, int]] = [{}]     # node -> {token: number of nouns through that edge}
		node_count = [0]
		for row in rows:
			node = 0
			node_count[0] += 1
			for tok in row.tolist():
				tok = int(tok)
				counts[node][tok] = counts[node].get(tok, 0) + 1
				if tok == 0:
					children[node].setdefault(0, -1)
					break
				nxt = children[node].get(tok)
				if nxt is None:
					nxt = len(children)
					children[node][tok] = nxt
					children.append({})
					counts.append({})
					node_count.append(0)
				node_count[nxt] += 1
				node = nxt
		start = np.zeros(len(children) + 1, dtype=np.int32)
		tok_l, next_l, lp_tgt, lp_tok = [], [], [], []
		for n, ch in enumerate(children):
			items = sorted(ch.items())
			start[n + 1] = start[n] + len(items)
			for t, nx in items:
				tok_l.append(t)
				next_l.append(nx)
				lp_tgt.append(math.log(counts[n][t] / node_count[n]))
				lp_tok.append(-math.log(len(items)))
		self.num_nodes, self.num_edges, self.num_targets = len(children), len(tok_l), len(rows)
		# per-target paths (generate_all, embedding_decoder.py:986-1041): node BEFORE column c and the CSR edge taken at column c; -1 after the END
		edge_of = [{t: int(start[n]) + k for k, (t, _) in enumerate(sorted(ch.items()))} for n, ch in enumerate(children)]
		cmax = rows.shape[1] if len(rows) else 0
		path_node = np.full((len(rows), cmax), -1, dtype=np.int32)
		path_edge = np.full((len(rows), cmax), -1, dtype=np.int32)
		for w, row in enumerate(rows):
			node = 0
			for c, tok in enumerate(row.tolist()):
				tok = int(tok)
				path_node[w, c], path_edge[w, c] = node, edge_of[node][tok]
				if tok == 0:
					break
				node = children[node][tok]
		self.path_node_host, self.path_edge_host = path_node, path_edge
		self._edge_of = edge_of
		self._children = children
		up = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(device)
		self.start, self.tok, self.next = up(start, np.int32), up(tok_l, np.int32), up(next_l, np.int32)
		self.logprior_target, self.logprior_token = up(lp_tgt, np.float32), up(lp_tok, np.float32)
		self.max_fanout = int((start[1:] - start[:-1]).max()) if len(children) else 0

	def prior_sums(self, targets: np.ndarray, valid: np.ndarray, per_token: bool) -> np.ndarray:
		"""sum over the valid columns of every row of `targets` (ANY nouns, not necessarily this trie's) of log P(token | prefix) under this trie; +inf when the
		row leaves the trie (the reference's nan_to_num(+inf) of log 0, embedding_decoder.py:1030)."""
		lp = (self.logprior_token if per_token else self.logprior_target).cpu().numpy()
		out = np.zeros(targets.shape[0], dtype=np.float32)
		for w in range(targets.shape[0]):
			node, total = 0, 0.0
			for c in range(targets.shape[1]):
				if not valid[w, c]:
					break
				tok = int(targets[w, c])
				e = self._edge_of[node].get(tok) if node >= 0 else None
				if e is None:
					total = float("inf")
					break
				total += float(lp[e])
				node = self._children[node][tok]
			out[w] = total
		return out


def trie_for(targets: torch.Tensor, device: torch.device) -> TokenTrie:
	"""The trie rides on the tensor OBJECT it was built from (guide / vocabulary tensors are built once per model load, infer.py:174, :687-710,
	and passed unchanged to every generate call); an in-place edit (version bump) or a different tensor -- even one recycled at the same
	address -- rebuilds it."""
	try:
		ver = targets._version
	except RuntimeError:  # inference tensors do not track versions
		ver = 0
	held = getattr(targets, "_novic_trie", None)
	if held is not None and held[0] == ver and held[1] == str(device):
		return held[2]
	trie = TokenTrie(targets, device)
	targets._novic_trie = (ver, str(device), trie)
	return trie

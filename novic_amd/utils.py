"""Small host utilities the infer/train surfaces need (subset of reference utils.py: AttrDict :386-401, flatten/unflatten :356-383,
dataclass_from_dict :334-344)."""
from __future__ import annotations

import dataclasses
from typing import Any


class AttrDict(dict):
	@classmethod
	def from_dict(cls, d: dict[str, Any]) -> "AttrDict":
		return cls({k: cls.from_dict(v) if isinstance(v, dict) else v for k, v in d.items()})

	def __getattr__(self, key: str) -> Any:
		try:
			return self[key]
		except KeyError as e:
			raise AttributeError(key) from e

	def __setattr__(self, key, value):
		self[key] = value


def flatten_dict(d: dict, parent_key=None) -> dict:
	out = {}
	for k, v in d.items():
		assert "." not in k
		nk = f"{parent_key}.{k}" if parent_key else k
		if isinstance(v, dict):
			out.update(flatten_dict(v, nk))
		else:
			out[nk] = v
	return out


def unflatten_dict(flat: dict) -> dict:
	out: dict = {}
	for key, v in flat.items():
		parts = key.split(".")
		cur = out
		for part in parts[:-1]:
			cur = cur.setdefault(part, {})
			if not isinstance(cur, dict):
				raise ValueError(f"Nesting conflict at '{part}' while inserting '{key}'")
		if parts[-1] in cur:
			raise ValueError(f"Duplicate key '{key}'")
		cur[parts[-1]] = v
	return out


def dataclass_from_dict(cls, state: dict[str, Any]):
	names = {f.name for f in dataclasses.fields(cls)}
	if names != set(state.keys()):
		raise ValueError(f"Cannot construct {cls.__qualname__}: missing {sorted(names - set(state))}, extra {sorted(set(state) - names)}")
	return cls(**state)

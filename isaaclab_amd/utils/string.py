"""Regex name resolution (host-side set-up code; mirrors ``isaaclab/utils/string.py:178-368``).

Term configs, actuator groups and action terms name joints / bodies by regular expression; these helpers
turn the expressions into index lists once, at construction time.  Nothing here runs per step.
"""

from __future__ import annotations

import re
from collections.abc import Sequence
from typing import Any


def _match(keys: Sequence[str], names: Sequence[str]):
    """For every name (in order) find the single key that fully matches it -> list of (name_idx, key_idx)."""
    hits: list[tuple[int, int]] = []
    matched_by: dict[int, str] = {}
    per_key: list[list[str]] = [[] for _ in keys]
    for ni, name in enumerate(names):
        for ki, key in enumerate(keys):
            if re.fullmatch(key, name):
                if ni in matched_by:
                    raise ValueError(f"Multiple matches for '{name}': '{matched_by[ni]}' and '{key}'!")
                matched_by[ni] = key
                hits.append((ni, ki))
                per_key[ki].append(name)
    return hits, per_key


def _check_all_matched(keys, per_key, names):
    if not all(per_key):
        msg = "\n" + "".join(f"\t{k}: {v}\n" for k, v in zip(keys, per_key)) + f"Available strings: {list(names)}\n"
        raise ValueError(
            f"Not all regular expressions are matched! Please check that the regular expressions are correct: {msg}"
        )


def resolve_matching_names(
    keys: str | Sequence[str], list_of_strings: Sequence[str], preserve_order: bool = False
) -> tuple[list[int], list[str]]:
    """string.py:178-271.  Result order: by position in ``list_of_strings``; with ``preserve_order`` by key first."""
    if isinstance(keys, str):
        keys = [keys]
    hits, per_key = _match(keys, list_of_strings)
    if preserve_order:
        hits = sorted(hits, key=lambda h: h[1])  # stable: names keep their relative order inside a key
    _check_all_matched(keys, per_key, list_of_strings)
    return [h[0] for h in hits], [list_of_strings[h[0]] for h in hits]


def resolve_matching_names_values(
    data: dict[str, Any], list_of_strings: Sequence[str], preserve_order: bool = False
) -> tuple[list[int], list[str], list[Any]]:
    """string.py:274-368.  Same as :func:`resolve_matching_names` with the dict value of the matching key attached."""
    if not isinstance(data, dict):
        raise TypeError(f"Input argument `data` should be a dictionary. Received: {data}")
    keys = list(data.keys())
    hits, per_key = _match(keys, list_of_strings)
    if preserve_order:
        hits = sorted(hits, key=lambda h: h[1])
    _check_all_matched(keys, per_key, list_of_strings)
    return [h[0] for h in hits], [list_of_strings[h[0]] for h in hits], [data[keys[h[1]]] for h in hits]

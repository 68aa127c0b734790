"""Topology loading and atom-selection resolution for the engine's own PDB topology.

Stands where the reference has ``pmarlo.utils.mdtraj`` (S/utils/mdtraj.py:21-92) and keeps its two public names
and their contract, since callers are written against it:

* ``load_mdtraj_topology(path)`` -> a topology object with ``select(expression)``;
* ``resolve_atom_selection(topo, selection, logger=, on_error=)`` -> list of int atom indices, or ``None``
  when no selection was asked for or the selection failed under a non-raising policy.

A selection is a failure when the expression does not parse, when an entry of an index list is not an integer
(ints, numpy ints and digit strings are), or when it names no atom at all.  ``on_error`` decides what a failure
does: ``"raise"`` (default) raises ``ValueError``, ``"warn"`` logs one warning on ``logger`` and yields ``None``,
``"ignore"`` yields ``None`` silently.  mdtraj itself is not needed: expressions go to
``pmarlo_amd.io.selection`` (the mdtraj selection grammar on ``pmarlo_amd.io.pdb.Topology``)."""

from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Literal, Sequence

from ..io.pdb import Topology, load_pdb

__all__ = ["load_mdtraj_topology", "resolve_atom_selection"]

_POLICIES = ("raise", "warn", "ignore")


def load_mdtraj_topology(topology: str | os.PathLike[str] | Path) -> Topology:
    """Topology of a PDB file."""
    return load_pdb(os.fspath(topology)).topology


@dataclass(frozen=True)
class _FailurePolicy:
    mode: str
    logger: logging.Logger | None

    def failed(self, reason: str, cause: Exception | None = None) -> None:
        """A selection could not be resolved: raise, warn or stay silent; returning means 'no selection'."""
        if self.mode == "raise":
            if cause is not None:
                raise cause
            raise ValueError(reason)
        if self.mode == "warn" and self.logger is not None:
            self.logger.warning("atom selection failed: %s", cause if cause is not None else reason)


def _indices_of(topo: Topology, selection) -> list[int]:
    """Atom indices named by an expression or listed one by one; raises ValueError / TypeError for what is neither."""
    if isinstance(selection, str):
        return [int(i) for i in topo.select(selection)]
    out = []
    for entry in selection:
        out.append(int(entry, 10) if isinstance(entry, str) else int(entry))
    return out


def resolve_atom_selection(topo: Topology, atom_selection: str | Sequence[int] | None, *,
                           logger: logging.Logger | None = None,
                           on_error: Literal["raise", "warn", "ignore"] = "raise") -> Sequence[int] | None:
    """Atom indices for ``atom_selection`` (expression, or sequence of indices) on ``topo``; see the module text."""
    if atom_selection is None:
        return None
    if on_error not in _POLICIES:
        raise ValueError("on_error must be 'raise', 'warn', or 'ignore'")
    policy = _FailurePolicy(on_error, logger)
    try:
        indices = _indices_of(topo, atom_selection)
    except (TypeError, ValueError) as exc:
        policy.failed("atom selection could not be read", exc)
        return None
    if not indices:
        policy.failed("atom selection produced no atoms")
        return None
    return indices

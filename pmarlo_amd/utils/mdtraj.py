"""Mirror of pmarlo.utils.mdtraj (S/utils/mdtraj.py): topology loading and atom-selection resolution, on the
engine's own PDB topology and selection parser instead of mdtraj's (same names, arguments and error behaviour:
``on_error`` = raise / warn / ignore, empty selections count as failures, sequences of ints or digit strings)."""

from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Callable, Literal, Sequence

from ..io.pdb import Topology, load_pdb

__all__ = ["load_mdtraj_topology", "resolve_atom_selection"]


def load_mdtraj_topology(topology: str | os.PathLike[str] | Path) -> Topology:
    """Load a topology from a PDB file (the reference: ``mdtraj.load_topology``, :21-24)."""
    return load_pdb(str(Path(topology))).topology


def _validate_on_error(on_error: str) -> None:
    if on_error not in {"raise", "warn", "ignore"}:
        raise ValueError("on_error must be 'raise', 'warn', or 'ignore'")


def _resolve_selection_from_string(topo: Topology, expression: str,
                                   handle_failure: Callable[[Exception | None], None]) -> Sequence[int] | None:
    try:
        selection = topo.select(expression)
    except (ValueError, TypeError) as exc:
        handle_failure(exc)
        return None
    if selection.size == 0:
        handle_failure(None)
        return None
    return [int(i) for i in selection]


def _resolve_selection_from_sequence(selection: Sequence[int | str],
                                     handle_failure: Callable[[Exception | None], None]) -> Sequence[int] | None:
    try:
        indices = [int(item, 10) if isinstance(item, str) else int(item) for item in selection]
    except (TypeError, ValueError) as exc:
        handle_failure(exc)
        return None
    if not indices:
        handle_failure(None)
        return None
    return indices


def resolve_atom_selection(topo: Topology, atom_selection: str | Sequence[int] | None, *,
                           logger: logging.Logger | None = None,
                           on_error: Literal["raise", "warn", "ignore"] = "raise") -> Sequence[int] | None:
    """Resolve an atom selection against ``topo`` (S/utils/mdtraj.py:67-92)."""
    if atom_selection is None:
        return None
    _validate_on_error(on_error)

    def _handle_failure(exc: Exception | None) -> None:
        if on_error == "raise":
            if exc is None:
                raise ValueError("atom selection produced no atoms")
            raise exc
        if on_error == "warn" and logger is not None:
            msg = "atom selection failed"
            logger.warning(msg if exc is None else f"{msg}: {exc}")

    if isinstance(atom_selection, str):
        return _resolve_selection_from_string(topo, atom_selection, _handle_failure)
    return _resolve_selection_from_sequence(atom_selection, _handle_failure)

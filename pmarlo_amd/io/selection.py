"""Atom-selection strings: the part of mdtraj's selection language that users of the reference pass.

The reference resolves ``atom_selection`` strings with ``mdtraj.Topology.select`` (S/utils/mdtraj.py:31-44,
S/markov_state_model/_loading.py:147-160, S/api/features.py:123-125, S/features/ramachandran.py:73).  mdtraj is not a
dependency of this engine; this module restates the published grammar for the forms that occur in practice:

  keywords     all / everything, none, protein, water, backbone, sidechain
  fields       name, resname, resid (0-based residue index), resSeq (PDB residue number), index (atom index),
               chainid (0-based chain index), element / symbol, serial
  values       one or several words or numbers (``name CA CB``), ranges ``resid 3 to 10``, comparisons
               ``index < 100``, ``resSeq >= 5``, ``name == CA``, ``element != H``
  logic        and / or / not (also && || !), parentheses

Residue classes follow mdtraj's published tables (mdtraj/core/residue_names.py, restated here because mdtraj is not
installed: `protein` is its _PROTEIN_RESIDUES -- the amino acids, their protonation / force-field variants (ASH, GLH,
HID / HIE / HIP, HSD / HSE / HSP, LYSH, CYX ...), the N- and C-terminal AMBER names (NALA, CALA ...) and the CAPS
ACE / NME / NAC / NH2 / NHE, so the reference's standard test system ACE-ALA-NME is protein throughout; `water` is its
_WATER_RESIDUES), backbone = protein atoms named N, CA, C, O, sidechain = protein atoms not named N, CA, C, O, HA, H.
Unknown words raise ValueError, as mdtraj does."""

from __future__ import annotations

import re

import numpy as np

__all__ = ["select", "PROTEIN_RESIDUES", "WATER_RESIDUES"]

_AMINO = "ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split()
PROTEIN_RESIDUES = frozenset(
    _AMINO
    # caps and unusual residues
    + "ACE NME NAC NH2 NHE CT3 AIB DALA HYP ORN PYRR MELEU MEVAL LSN QLN".split()
    # protonation states and force-field spellings
    + """ARGN ASN1 ASP1 ASPH ASH CYS1 CYS2 CYSH CYM CYX GLUH GLH HID HIE HIP HIS1 HISA HISB HISH HISD HISE HISP
         HSD HSE HSP LYSH LYN""".split()
    # AMBER names of the chain ends
    + ["N" + r for r in _AMINO + ["HID", "HIE", "HIP", "CYX"]] + ["C" + r for r in _AMINO + ["HID", "HIE", "HIP", "CYX"]]
    # kept from earlier rounds of this module (ambiguous / rare codes PDB files carry)
    + "ASX GLX PYL SEC UNK".split())
WATER_RESIDUES = frozenset("H2O HHO OHH HOH OH2 SOL WAT TIP TIP2 TIP3 TIP4 W".split())
_BACKBONE = frozenset(("N", "CA", "C", "O"))
_NOT_SIDECHAIN = frozenset(("N", "CA", "C", "O", "HA", "H"))

_TOKEN = re.compile(r"\s*(\(|\)|<=|>=|==|!=|<|>|&&|\|\||!|[^\s()<>=!&|]+)")
_FIELDS = {"name": "name", "resname": "resname", "resn": "resname", "resid": "resid", "residue": "resSeq",
           "resseq": "resSeq", "resSeq": "resSeq", "index": "index", "chainid": "chainid", "element": "element",
           "symbol": "element", "type": "element", "serial": "serial"}
_KEYWORDS = ("all", "everything", "none", "protein", "is_protein", "water", "waters", "is_water", "backbone",
             "is_backbone", "sidechain", "is_sidechain")
_CMP = ("<", "<=", ">", ">=", "==", "!=")
_LOGIC = ("and", "or", "not", "&&", "||", "!")


def _tokens(text: str) -> list[str]:
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"cannot parse selection at {text[pos:]!r}")
        out.append(m.group(1))
        pos = m.end()
    return out


class _Parser:
    def __init__(self, topo, text: str):
        self.t = topo
        self.tok = _tokens(text)
        self.i = 0
        self.n = topo.n_atoms

    def peek(self):
        return self.tok[self.i] if self.i < len(self.tok) else None

    def take(self):
        tk = self.peek()
        self.i += 1
        return tk

    # expr := and_expr ('or' and_expr)*
    def expr(self) -> np.ndarray:
        m = self.and_expr()
        while self.peek() in ("or", "||"):
            self.take()
            m = m | self.and_expr()
        return m

    def and_expr(self) -> np.ndarray:
        m = self.not_expr()
        while self.peek() in ("and", "&&"):
            self.take()
            m = m & self.not_expr()
        return m

    def not_expr(self) -> np.ndarray:
        if self.peek() in ("not", "!"):
            self.take()
            return ~self.not_expr()
        return self.atom()

    def atom(self) -> np.ndarray:
        tk = self.take()
        if tk is None:
            raise ValueError("selection ends unexpectedly")
        if tk == "(":
            m = self.expr()
            if self.take() != ")":
                raise ValueError("missing ')' in selection")
            return m
        if tk in _KEYWORDS:
            return self.keyword(tk)
        if tk in _FIELDS:
            return self.field(_FIELDS[tk])
        raise ValueError(f"unknown selection word {tk!r}")

    def keyword(self, kw: str) -> np.ndarray:
        t = self.t
        names = np.asarray(t.atom_names)
        resn = np.asarray([r.upper() for r in t.res_names])
        prot = np.isin(resn, list(PROTEIN_RESIDUES))
        if kw in ("all", "everything"):
            return np.ones(self.n, bool)
        if kw == "none":
            return np.zeros(self.n, bool)
        if kw in ("protein", "is_protein"):
            return prot
        if kw in ("water", "waters", "is_water"):
            return np.isin(resn, list(WATER_RESIDUES))
        if kw in ("backbone", "is_backbone"):
            return prot & np.isin(names, list(_BACKBONE))
        return prot & ~np.isin(names, list(_NOT_SIDECHAIN))     # sidechain

    def _column(self, field: str):
        t = self.t
        if field == "name":
            return np.asarray(t.atom_names), False
        if field == "resname":
            return np.asarray(t.res_names), False
        if field == "element":
            return np.asarray(t.elements), False
        if field == "resid":
            return np.asarray(t.res_index, np.int64), True
        if field == "resSeq":
            return np.asarray(t.res_seq, np.int64), True
        if field == "index":
            return np.arange(self.n, dtype=np.int64), True
        if field == "serial":
            return np.asarray(t.serials, np.int64), True
        return np.asarray(t.chain_index, np.int64), True       # chainid

    def field(self, field: str) -> np.ndarray:
        col, numeric = self._column(field)

        def conv(v: str):
            if not numeric:
                return v.strip("'\"")
            try:
                return int(v)
            except ValueError as exc:
                raise ValueError(f"{field} needs integer values, got {v!r}") from exc

        tk = self.peek()
        if tk in _CMP:
            op = self.take()
            v = conv(self.take() or "")
            if not numeric and op not in ("==", "!="):
                raise ValueError(f"{field} supports only == and != comparisons")
            return {"<": col < v, "<=": col <= v, ">": col > v, ">=": col >= v, "==": col == v, "!=": col != v}[op] \
                if numeric else ((col == v) if op == "==" else (col != v))
        vals = []
        mask = np.zeros(self.n, bool)
        while self.peek() is not None and self.peek() not in _LOGIC and self.peek() not in ("(", ")") \
                and self.peek() not in _KEYWORDS and self.peek() not in _FIELDS:
            tk = self.take()
            if tk == "to":
                if not numeric or not vals:
                    raise ValueError(f"'to' needs a numeric field and a start value ({field})")
                hi = conv(self.take() or "")
                lo = vals.pop()
                mask |= (col >= lo) & (col <= hi)
            else:
                vals.append(conv(tk))
        if not vals and not mask.any() and self.i > 0 and self.tok[self.i - 1] != "to":
            if not vals:
                raise ValueError(f"{field} needs at least one value")
        for v in vals:
            mask |= col == v
        return mask


def select(topology, query: str) -> np.ndarray:
    """Indices (ascending) of the atoms of `topology` matched by the mdtraj-style selection `query`."""
    if not isinstance(query, str) or not query.strip():
        raise ValueError("empty selection")
    p = _Parser(topology, query)
    mask = p.expr()
    if p.peek() is not None:
        raise ValueError(f"unexpected {p.peek()!r} in selection {query!r}")
    return np.flatnonzero(mask).astype(int)

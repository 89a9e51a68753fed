"""Proposal generators: same names, keyword sets and constructor errors as
/root/reference/src/proposals.jl.  The jump itself runs inside the fused update kernel
(csrc/kernels.hip k_update); these objects only carry the parameters."""
from __future__ import annotations

import math

from . import _lib
from ._lib import SABCError


class Proposal:
    def descriptor(self):
        raise NotImplementedError


class RandomWalk(Proposal):
    """RandomWalk(; β=0.8, n_para) -- proposals.jl:24-36.  Σ is learned from the population
    (proposals.jl:46-48,58-60); `Σ` mirrors the reference's field and is refreshed after updates."""

    def __init__(self, *, β=0.8, n_para):
        if not (0 < β <= 1):
            raise SABCError(-6, "Mixing parameter `β` must be between zero and one.")   # proposals.jl:30
        self.β = float(β)
        self.n_para = int(n_para)
        self.Σ = -1.0 if n_para == 1 else [[-1.0] * n_para for _ in range(n_para)]      # proposals.jl:32,34

    def descriptor(self):
        return (_lib.PROP_RANDOMWALK, self.β, 0.0)

    def __repr__(self):
        return f"RandomWalk(β={self.β}, n_para={self.n_para})"


class DifferentialEvolution(Proposal):
    """DifferentialEvolution(; n_para | γ0, σ_gamma=1e-5) -- proposals.jl:85-99.
    Keyword-only like the reference (positional use is a MethodError there, a TypeError here)."""

    def __init__(self, *, γ0=None, n_para=None, σ_gamma=1e-5):
        if γ0 is not None and n_para is None:
            self.γ0 = float(γ0)
        elif n_para is not None and γ0 is None:
            self.γ0 = 2.38 / math.sqrt(2 * n_para)                                      # proposals.jl:93
        else:
            raise ValueError("Provide either `γ0` or `n_para`, not both.")              # ArgumentError, :96
        self.σ_gamma = float(σ_gamma)

    def descriptor(self):
        return (_lib.PROP_DIFFEVO, self.γ0, self.σ_gamma)

    def __repr__(self):
        return f"DifferentialEvolution(γ0={self.γ0}, σ_gamma={self.σ_gamma})"


class StretchMove(Proposal):
    """StretchMove(; a=2) -- proposals.jl:132-135."""

    def __init__(self, a=2.0, **kw):
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        self.a = float(a)

    def descriptor(self):
        return (_lib.PROP_STRETCH, self.a, 0.0)

    def __repr__(self):
        return f"StretchMove(a={self.a})"

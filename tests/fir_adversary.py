"""Symbols chosen AGAINST the matrix-core FIR (fir_fused.hip): a float64 restatement of what k_mix_fir_mm leaves out of its
outer K-steps, built from the tap operand the library multiplies by (atsc_hip_get_table id 8) and the kernel's own split
of the oscillator (fir_fft.hip: k_osc_table, planes form), used to pick for every input of a target output the level that
pushes the omitted terms one way.  Test infrastructure only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
import fir_error_budget as feb  # noqa: E402

LEVELS = 2.0 * np.arange(8) - 7.0 + 1.25   # signal.h:13-18


def f16(v):
    return np.asarray(v, np.float64).astype(np.float16).astype(np.float64)


def z_pieces(t_comp):
    """One oscillator component (fp32) at some positions -> for each of the 8 levels: the exact z' = level * T * 4096 and the
    kernel's pieces Zhi, Zlo and the once-rounded fl16(Zhi + Zlo) (mix_pair / the outer K-steps' operand)."""
    tp = (np.asarray(t_comp, np.float32) * np.float32(4096.0))
    th = tp.astype(np.float16)
    tl = (tp - th.astype(np.float32)).astype(np.float16)
    th, tl, tp = th.astype(np.float64), tl.astype(np.float64), tp.astype(np.float64)
    lv = LEVELS[:, None]
    exact = lv * tp[None, :]
    prod = lv * th[None, :]
    zhi = f16(prod)
    zlo = f16(lv * tl[None, :] + (prod - zhi))
    return exact, zhi, zlo, f16(zhi + zlo)


def omitted(phase, p, t_window, operand):
    """For output row p of a tile and the 144 inputs of its window (oscillator component values t_window): what the
    kernel's result lacks, per candidate level and input, in units of full scale; and the mask of inputs it matters for."""
    hi, lo, true = (v[phase] for v in operand)            # [K-step 9][row 32][k 16]
    a_true = true[:, p, :].reshape(144)
    a_hi = hi[:, p, :].reshape(144)
    ks = np.arange(144) // 16
    exact, zhi, zlo, zonce = z_pieces(t_window)
    outer = np.isin(ks, (0, 8))
    near = np.isin(ks, (1, 7))
    used = np.where(outer[None, :], zonce, zhi + zlo)
    err = (a_true[None, :] * exact - a_hi[None, :] * used) / 4096.0
    touched = (outer | near) & (a_true != 0)
    return err, touched


def choose(sym, c, tile, p, comp, t_field, phase, operand, sign):
    """Overwrites the symbols of channel row c that feed the outer K-steps of output (tile, p) so that the omitted terms
    of component comp add up with sign `sign`.  t_field: [padded positions][2] fp32.  Returns the error aimed for."""
    n = sym.shape[1]
    pos = 32 * tile + np.arange(144)
    err, touched = omitted(phase, p, t_field[pos, comp], operand)
    pick = np.argmax(err * sign, axis=0)
    inside = pos < n                       # positions behind the field are the copy of its first segment: left alone
    sel = touched & inside
    sym[c, pos[sel]] = pick[sel]
    return float((err[pick, np.arange(144)] * sel).sum())


def operand(handle=None):
    return feb.fused_operand(handle)

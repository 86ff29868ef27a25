"""Time-splitting ONE transport stream at field boundaries across ranks (BASELINE config 5).

The reference is strictly field-serial (src/atsc/atsc.cpp:59-90).  Everything a span of fields needs
from the past is small:

  * interleaver history  -- a function of the previous field's packets only, so a rank re-derives it
                            from a one-field TS halo (no message);
  * field parity, oscillator phase -- functions of the absolute field index (recomputed locally by
                            atsc_hip_import_state with the reference's own fp32 recursion);
  * trellis/precoder state -- GF(2)-linear: state_in(rank r) = XOR of the span parities of ranks < r.
                            This is the ONE exchange step: an all-gather of 16 bytes per rank and
                            channel (RCCL over xGMI on GPUs, gloo in the CPU tests), then a local XOR fold;
  * reserved sync symbols -- follow from the carries and the halo field's last interleaved group.

No data-path tensor ever crosses GPUs; the message is latency-bound.

A handle may carry several streams (`channels`): all of them are split at the same field boundaries and their
parities travel in the same message.  The forms of the mixer + FIR differ in the last bits of the IQ, so the spans
join bit-identically only if every rank runs the same form: `check_uniform` compares (channels, resolved form,
sample format) across the ranks and raises when they differ.
"""
import torch

import atsc_amd

PARITY_BYTES = 16  # per channel: 3 little-endian words (encoders 0-3, 4-7, 8-11) + 1 pad word


def _gather(t, group):
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    buf = t if backend == "nccl" else t.cpu()
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return gathered


def exclusive_xor_over_ranks(span_parity, group=None):
    """All-gather every rank's span parity and return the XOR of the parities of all EARLIER ranks.

    span_parity: int32 tensor [4*channels] (the 16-byte-per-channel format of atsc_hip_encode_begin).
    Works with the nccl (RCCL) backend on device tensors and with gloo (tensors are staged through the
    host).  With no initialised process group (single process) it returns zeros.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return torch.zeros_like(span_parity)
    rank = dist.get_rank(group)
    gathered = _gather(span_parity, group)
    carry = torch.zeros_like(gathered[0])
    for q in range(rank):
        carry ^= gathered[q]
    return carry.to(span_parity.device)


def check_uniform(values, group=None, device="cpu"):
    """Raise unless every rank holds the same tuple of ints (channels, FIR form, sample format ...)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    mine = torch.tensor(list(values), dtype=torch.int32, device=device)
    for q, other in enumerate(_gather(mine, group)):
        if other.cpu().tolist() != list(values):
            raise atsc_amd.AtscHipError(
                f"time-split ranks disagree: rank {dist.get_rank(group)} has (channels, fir_form, iq_format, ...) = "
                f"{list(values)}, rank {q} has {other.cpu().tolist()}; the spans would not join bit-identically")


class TimeSplitEncoder:
    """One rank's share of a time-split: fields [first_field, first_field + fields) of `channels` streams."""

    def __init__(self, fields, device=0, channels=1, fir_form=atsc_amd.FIR_AUTO, iq_format=atsc_amd.IQ_CF32, upsample=1):
        self.fields = fields
        self.device = device
        self.channels = channels
        self.iq_format = iq_format
        self.enc = atsc_amd.Encoder(channels=channels, max_fields=fields, device=device)
        self.enc.set_fir_form(fir_form)
        self.enc.set_iq_format(iq_format)
        self.enc.set_upsample(upsample)      # 2: d_iq holds 2 x 260416 samples per field (a field is self-contained in both forms)
        self._checked = False

    def prepare(self, first_field, halo_field_ts=None):
        """Set up the history entering field `first_field`.  halo_field_ts: device u8 tensor with the 312
        packets of field first_field-1 of every channel, [channels][312*188] (None only for first_field == 0)."""
        if first_field == 0:
            return
        assert halo_field_ts is not None and halo_field_ts.numel() == self.channels * atsc_amd.FIELD_TS_BYTES
        stream = torch.cuda.current_stream().cuda_stream
        # warm-up field: leaves the right interleaver tail and last interleaved group in the handle
        self.enc.encode_batch(halo_field_ts.data_ptr(), 1, None, None, stream)
        # trellis state comes from the exchange, not from the halo; field parity + oscillator phase of the span start.
        # All channels at once, on the device, in stream order (no host round trip)
        self.enc.seek(first_field, stream)

    def encode(self, d_ts, d_sym, d_iq, group=None, exchange=exclusive_xor_over_ranks):
        """begin -> exchange span parities -> finish.  d_ts/d_sym/d_iq: device tensors of this span
        ([channels][fields][...])."""
        if not self._checked:
            check_uniform((self.channels, self.enc.fir_form(), self.iq_format, self.enc.upsample()), group,
                          device=d_ts.device if _is_nccl(group) else "cpu")
            self._checked = True
        stream = torch.cuda.current_stream().cuda_stream
        par = torch.zeros(4 * self.channels, dtype=torch.int32, device=d_ts.device)
        self.enc.encode_begin(d_ts.data_ptr(), self.fields, par.data_ptr(), stream)
        carry = exchange(par, group).contiguous()
        self.enc.encode_finish(carry.data_ptr(), d_sym.data_ptr() if d_sym is not None else None,
                               d_iq.data_ptr() if d_iq is not None else None, stream)
        self._keep = (par, carry)  # keep the tensors alive until the stream has consumed them
        return par


def _is_nccl(group):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"

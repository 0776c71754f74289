"""Synthetic input batches for the BASELINE.json configs.

Measurement tooling shared by bench.py and the full-size GPU tests; seeded and
generated on the device that will scan them (nothing is shipped).  All batches
are fixed pitch: uint8 tensor [n, L], one text per row.
"""
from __future__ import annotations

import torch

_LOWER0, _DIGIT0 = 97, 48


def _gen(seed: int, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _token_stream(nbytes: int, g, device) -> torch.Tensor:
    """Space separated words [a-z]{1..12}[0-9]{0..6} (about half with digits)."""
    # average token = 6.5 letters + 1.75 digits + 1 space ~ 9.25 bytes
    ntok = int(nbytes / 8.5) + 16
    nlet = torch.randint(1, 13, (ntok,), generator=g, device=device)
    ndig = torch.randint(1, 7, (ntok,), generator=g, device=device)
    ndig = torch.where(torch.rand(ntok, generator=g, device=device) < 0.5, ndig, torch.zeros_like(ndig))
    tlen = nlet + ndig + 1
    ends = torch.cumsum(tlen, 0)
    total = int(ends[-1])
    assert total >= nbytes
    tok = torch.repeat_interleave(torch.arange(ntok, device=device), tlen)[:nbytes]
    starts = ends - tlen
    off = torch.arange(nbytes, device=device) - starts[tok]
    letters = torch.randint(0, 26, (nbytes,), generator=g, device=device) + _LOWER0
    digits = torch.randint(0, 10, (nbytes,), generator=g, device=device) + _DIGIT0
    out = torch.where(off < nlet[tok], letters,
                      torch.where(off < (nlet + ndig)[tok], digits, torch.full_like(letters, 32)))
    return out.to(torch.uint8)


def make_c2_batch(n: int, L: int = 1024, seed: int = 20260102, device="cuda") -> torch.Tensor:
    """Config 2 ([a-z]+\\d+): 40 % full, 30 % tokens, 20 % noise, 10 % adversarial.

      full         [a-z]{k}[0-9]{L-k}, k ~ U[1, L-1]   (one match spanning the text)
      tokens       space separated [a-z]{1..12}[0-9]{0..6} words (many matches)
      noise        i.i.d. uniform over the 95 printable bytes (early death, few hits)
      adversarial  L-1 lowercase bytes + '!' (worst case for restart-per-position search)
    """
    g = _gen(seed, device)
    kind_r = torch.rand(n, generator=g, device=device)
    kind = (kind_r >= 0.4).to(torch.int64) + (kind_r >= 0.7) + (kind_r >= 0.9)
    out = torch.empty((n, L), dtype=torch.uint8, device=device)
    # process in row blocks to bound temporary memory
    step = max(1, min(n, (64 << 20) // max(L, 1)))
    col = torch.arange(L, device=device)
    for a in range(0, n, step):
        b = min(n, a + step)
        m = b - a
        k = kind[a:b]
        letters = (torch.randint(0, 26, (m, L), generator=g, device=device) + _LOWER0).to(torch.uint8)
        digits = (torch.randint(0, 10, (m, L), generator=g, device=device) + _DIGIT0).to(torch.uint8)
        split = torch.randint(1, max(L, 2), (m, 1), generator=g, device=device)
        full = torch.where(col[None, :] < split, letters, digits)
        noise = (torch.randint(0, 95, (m, L), generator=g, device=device) + 32).to(torch.uint8)
        adv = letters.clone()
        adv[:, L - 1] = 33  # '!'
        tokens = _token_stream(m * L, g, device).reshape(m, L)
        blk = torch.where((k == 0)[:, None], full,
                          torch.where((k == 1)[:, None], tokens,
                                      torch.where((k == 2)[:, None], noise, adv)))
        out[a:b] = blk
    return out


def make_digits_batch(n: int, L: int = 256, seed: int = 20260103, device="cuda") -> torch.Tensor:
    """Config 3 (\\d+ findall): bytes with P(digit) ~ 0.08 in runs (a run continues with
    probability 0.6, i.e. run length ~ Geom(0.4)), about 4 digit runs per 256-byte text."""
    g = _gen(seed, device)
    out = torch.empty((n, L), dtype=torch.uint8, device=device)
    step = max(1, min(n, (128 << 20) // max(L, 1)))
    p_stay, p_on = 0.6, 0.017
    for a in range(0, n, step):
        b = min(n, a + step)
        m = b - a
        r = torch.rand((m, L), generator=g, device=device)
        on = torch.zeros((m, L), dtype=torch.bool, device=device)
        prev = torch.zeros(m, dtype=torch.bool, device=device)
        for j in range(L):
            cur = torch.where(prev, r[:, j] < p_stay, r[:, j] < p_on)
            on[:, j] = cur
            prev = cur
        digits = (torch.randint(0, 10, (m, L), generator=g, device=device) + _DIGIT0).to(torch.uint8)
        other = (torch.randint(0, 26, (m, L), generator=g, device=device) + _LOWER0).to(torch.uint8)
        sp = torch.rand((m, L), generator=g, device=device) < 0.15
        other = torch.where(sp, torch.full_like(other, 32), other)
        out[a:b] = torch.where(on, digits, other)
    return out


def make_phone_batch(n: int, L: int = 1024, seed: int = 20260104, device="cuda") -> torch.Tensor:
    """Config 4 ((\\d{3})(\\d{3})(\\d{4})): 'Call 6502530000 or 4155551234 today. ' style
    text with random digits; 25 % of the texts hold only 9-digit runs (near misses) and
    5 % hold 11..25-digit runs (leftmost start inside long runs)."""
    g = _gen(seed, device)
    tmpl = torch.tensor(list(b"Call DDDDDDDDDD or DDDDDDDDDD today. "), dtype=torch.uint8, device=device)
    reps = (L + tmpl.numel() - 1) // tmpl.numel()
    row = tmpl.repeat(reps)[:L]
    is_d = (row == ord("D"))[None, :]
    tpos = torch.arange(L, device=device) % tmpl.numel()
    tenth = ((tpos == 14) | (tpos == 28))[None, :]        # 10th digit of each number
    sep = ((tpos >= 15) & (tpos <= 18))[None, :]           # " or " between the two numbers
    out = torch.empty((n, L), dtype=torch.uint8, device=device)
    step = max(1, min(n, (64 << 20) // max(L, 1)))
    for a in range(0, n, step):
        b = min(n, a + step)
        m = b - a
        digits = (torch.randint(0, 10, (m, L), generator=g, device=device) + _DIGIT0).to(torch.uint8)
        kind = torch.rand(m, generator=g, device=device)
        near = (kind < 0.25)[:, None] & tenth                 # near misses: 9-digit runs only
        longr = ((kind >= 0.25) & (kind < 0.30))[:, None] & sep  # 24-digit runs
        blk = torch.where(is_d | longr, digits, row[None, :].expand(m, L))
        out[a:b] = torch.where(near, torch.full_like(blk, ord("x")), blk)
    return out


def make_alt_batch(n: int, L: int = 4096, seed: int = 20260105, device="cuda") -> torch.Tensor:
    """Config 5 ((x|y|foo|bar)+): i.i.d. over {x,y,f,o,b,a,r,' '}, weights chosen so that
    foo / bar complete about 30 % of the time."""
    g = _gen(seed, device)
    sym = torch.tensor(list(b"xyfobar "), dtype=torch.uint8, device=device)
    w = torch.tensor([0.08, 0.08, 0.14, 0.26, 0.12, 0.12, 0.12, 0.08], device=device)
    cdf = torch.cumsum(w, 0)
    out = torch.empty((n, L), dtype=torch.uint8, device=device)
    step = max(1, min(n, (256 << 20) // max(L, 1)))  # bound temporaries to ~1 GiB of floats
    for a in range(0, n, step):
        b = min(n, a + step)
        r = torch.rand((b - a, L), generator=g, device=device)
        idx = torch.bucketize(r, cdf[:-1], right=True)
        out[a:b] = sym[idx]
    return out


def to_ragged(batch: torch.Tensor, min_len: int = 0, seed: int = 20260110):
    """CSR form of a fixed-pitch batch: row i is cut to a length drawn uniformly from
    [min_len, L] and the rows are packed back to back.  Returns (data u8[total], offsets i64[n+1])."""
    n, L = batch.shape
    g = _gen(seed, batch.device)
    lens = torch.randint(min_len, L + 1, (n,), generator=g, device=batch.device)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=batch.device)
    offsets[1:] = torch.cumsum(lens, 0)
    data = torch.empty(int(offsets[-1].item()), dtype=torch.uint8, device=batch.device)
    col = torch.arange(L, device=batch.device)
    step = max(1, (256 << 20) // max(L, 1))
    for a in range(0, n, step):
        b = min(n, a + step)
        keep = col[None, :] < lens[a:b, None]
        data[int(offsets[a].item()):int(offsets[b].item())] = batch[a:b][keep]
    return data, offsets

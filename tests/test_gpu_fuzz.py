"""Randomised link configurations through the batched engine against the CPU oracle (GPU).

Every seed draws one link -- carrier, modulation, layers / antennas, code rate and base graph, DMRS, channel model, time- or
frequency-domain, LS or perfect CSI -- and two slots of it in parity mode (transport blocks and noise draws as data).  Checked:
  (i)   the engine's LLRs equal oracle.link.run_slot's (the reference's loop body, PDSCH-BLER.ipynb cell 2) to 1e-9 of the slot's scale;
  (ii)  the decoder on the ORACLE's LLRs -- separate stages and, where the configuration has one, the fused entry -- gives the oracle's CRC
        verdicts and hard bits exactly (float64, ldpc.py:1330-1619);
  (iii) the throughput path of the same link (fused stages) ends on the verdicts of the stage-by-stage path.
The collected test runs a fixed set of seeds; `python tests/test_gpu_fuzz.py FIRST LAST` sweeps a range and prints one line per seed.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

pytestmark = pytest.mark.gpu

MODS = ['QPSK', '16QAM', '64QAM', '256QAM']


def draw(seed):
    r = np.random.default_rng(10_000 + seed)
    spacing = int(r.choice([15, 30]))
    num_rbs = int(r.integers(4, 53 if spacing == 15 else 52))
    n_ant = int(r.choice([1, 2, 4]))
    layers = int(r.choice([l for l in (1, 2, 4) if l <= n_ant]))
    cdl = bool(r.random() < 0.6) and n_ant >= 2
    prof = str(r.choice(list('ABCDE')))
    ds = float(r.choice([10, 30, 100, 300]))
    dop = float(r.choice([0, 5, 70]))
    if cdl:
        panel = {2: [1, 1], 4: [1, 2]}[n_ant]
        chan = ('cdl', prof, ds, dop, panel, panel)
    else:
        chan = ('tdl', prof, ds, dop, n_ant, n_ant)
    return dict(seed=seed, numRbs=num_rbs, spacing=spacing, mod=str(r.choice(MODS)), layers=layers,
                dm=dict(configType=int(r.choice([1, 2])), additionalPos=int(r.integers(0, 3))), chan=chan, slot0=0,
                rate=float(r.uniform(0.15, 0.92)), bg=int(r.choice([1, 2])), numIter=int(r.integers(3, 9)),
                freqDomain=bool(r.random() < 0.35), perfect=bool(r.random() < 0.35), snr=float(r.uniform(-2, 32)))


def deinterleave_per_code_block(llr, C, G, nl, qm):
    """Symbol-major LLRs -> the layout the demapper writes for the fused decoder entry: inside code block r (E_r values, ldpc.py:846-856)
    value q of symbol s sits at q * (E_r / Qm) + s (ldpc.py:1390-1397 done once, by position)."""
    f = nl * qm
    gb = -(-G // f)
    e_small, n_small = (gb // C) * f, C - gb % C
    out, off = np.empty_like(llr), 0
    for r in range(C):
        e = e_small if r < n_small else e_small + f
        out[off:off + e] = llr[off:off + e].reshape(e // qm, qm).T.reshape(-1)
        off += e
    assert off == len(llr)
    return out


def run_case(seed, n=2):
    import torch
    import neoradium_amd as nr
    from neoradium_amd import ops
    from neoradium_amd._dev import D
    from oracle import link as olink
    from test_gpu_classes import _slot
    c = draw(seed)
    car, bwp, p, ch = _slot(nr, c)
    link = nr.PdschLink(p, ch, c['rate'], baseGraphNo=c['bg'], numIter=c['numIter'], freqDomain=c['freqDomain'],
                        chanEst="Perfect" if c['perfect'] else "LS", decoder="f64")
    rng = np.random.default_rng(seed)
    tb = rng.integers(0, 2, (n, link.tbs)).astype(np.uint8)
    shape = (link.nr, link.L, link.K) if c['freqDomain'] else (link.nr, bwp.getSlotLen(0) + link.max_delay)
    z = rng.standard_normal((n,) + shape + (2,))
    zc = z[..., 0] + 1j * z[..., 1]
    counters, det = link.run(0, n, c['snr'], tb_bits=torch.from_numpy(tb), noise=D(zc), details=True)
    assert len(det) == 1
    d = det[0][1]
    _, dv = link.run(0, n, c['snr'], tb_bits=torch.from_numpy(tb), noise=D(zc), details="verdicts")
    assert torch.equal(dv[0][1]['cb_ok'].reshape(-1), d['cb_ok'].reshape(-1)), "throughput path and stage-by-stage path disagree"
    cw = link.cw[0]
    st = olink.static_from_link(link)
    info = dict(c, tbs=link.tbs, C=link.cfg.C, Zc=link.cfg.Zc, rows=cw['rows'], blocks_ok=0, blocks=0, llr_err=0.0)
    for s in range(n):
        ref = olink.run_slot(st, s, c['snr'], tb[s].astype(np.int8), zc[s], F=d['F'][s].cpu().numpy())
        got = d['llr'][s].cpu().numpy()
        err = float(np.abs(got - ref['llr']).max() / np.abs(ref['llr']).max())
        info['llr_err'] = max(info['llr_err'], err)
        assert err <= 1e-9, ("LLRs", err)
        nb = len(ref['tb_out'])
        want = ref['tb_out'].astype(np.uint8)
        llr_o = D(ref['llr'][None])
        rr = ops.ldpc_rate_recover(llr_o, cw['cfg'], cw['nl'], cw['qm'])
        dec = ops.ldpc_decode(rr, cw['cfg'], link.numIter, rows=cw['rows'])
        tb_o, cb_ok, _ = ops.ldpc_crc_merge(dec, cw['cfg'], want_tb_crc=False)
        assert np.array_equal(cb_ok[0].cpu().numpy().astype(bool), ref['crc']), "decoder on the oracle's LLRs: CRC verdicts"
        assert np.array_equal(tb_o[0].cpu().numpy()[:nb], want), "decoder on the oracle's LLRs: hard bits"
        if ops.ldpc_fused_supported(cw['cfg'], cw['nl'], cw['qm'], cw['G'], cw['rows']):
            llr_cb = D(deinterleave_per_code_block(ref['llr'], link.cfg.C, cw['G'], cw['nl'], cw['qm'])[None])
            ft, fo = ops.ldpc_recover_decode_merge(llr_cb, cw['cfg'], cw['nl'], cw['qm'], link.numIter, rows=cw['rows'])[:2]
            assert np.array_equal(fo[0].cpu().numpy().astype(bool), ref['crc']) and np.array_equal(ft[0].cpu().numpy()[:nb], want), "fused entry"
            info['fused'] = True
        info['blocks_ok'] += int(ref['crc'].sum())
        info['blocks'] += len(ref['crc'])
    return info


@pytest.mark.parametrize("seed", list(range(14)))
def test_random_link_against_the_oracle(dev, seed):
    run_case(seed)


def run_fused_case(seed):
    """The fused decoder entry (nrx_ldpc_recover_decode_merge_f64: BG1, Zc 384, <= 15 rows) at a random geometry -- bits per code block
    anywhere between 4 and 15 rows' worth, unequal E_r, any Qm x layers -- against the oracle's rate recovery + decode + CRC (ldpc.py:1330-1619)
    on noisy LLRs: covers the last-layer skip of the waves (NRX_DEC3_SKIPZ) and the copies that leave whole layers out (MODE bit 3)."""
    import torch
    from neoradium_amd import ops, _lib
    from neoradium_amd._dev import D
    from oracle import coding as oc
    r = np.random.default_rng(77_000 + seed)
    C = int(r.integers(2, 7))
    cfg = None
    for _ in range(50):
        tbs = int(r.integers(8000 * C - 7000, 8424 * C - 24))
        cfg = _lib.ldpc_config(1, tbs + 24)
        if cfg.Zc == 384 and cfg.C == C:
            break
    assert cfg.Zc == 384 and cfg.C == C
    qm, nl = int(r.choice([2, 4, 6, 8])), int(r.choice([1, 2, 3, 4]))
    f = qm * nl
    e_lo = cfg.K - 2 * 384 - cfg.F + 384            # at least one parity column
    e = int(r.integers(e_lo, 13300)) // f * f
    g_extra = int(r.integers(0, C))
    G = C * e + g_extra * f
    lens = _lib.ldpc_cb_lens(G, C, nl, qm)
    rows = ops.ldpc_active_rows(cfg, max(lens))
    if rows > 15 or not ops.ldpc_fused_supported(cfg, nl, qm, G, rows):
        return dict(seed=seed, skipped=True, rows=rows)
    n_tb, n_it = 2, int(r.integers(3, 10))
    tb = torch.from_numpy(r.integers(0, 2, (n_tb, tbs)).astype(np.uint8)).to(torch.device('cuda:0'))
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm).cpu().numpy().astype(np.float64)
    sig = r.uniform(0.45, 0.9, (n_tb, 1))
    llr = (2 / sig ** 2) * ((1 - 2 * bits) + sig * r.standard_normal(bits.shape))
    llr[r.random(llr.shape) < 0.001] = 0.0
    xd = np.stack([deinterleave_per_code_block(x, C, G, nl, qm) for x in llr])
    tb_out, ok = ops.ldpc_recover_decode_merge(D(xd), cfg, nl, qm, n_it, rows=rows)
    pp = oc.LdpcParams(1, tbs + 24)
    assert (pp.C, pp.Zc, pp.F) == (cfg.C, cfg.Zc, cfg.F)
    for t in range(n_tb):
        rr, _ = oc.rate_recover(llr[t], pp, nl, qm)
        dec = oc.decode(rr, 1, pp.iLS, pp.Zc, n_it)
        out, crc = oc.crc_check_and_merge(dec, pp)
        assert np.array_equal(ok[t].cpu().numpy().astype(bool), crc), (seed, "CRC verdicts")
        assert np.array_equal(tb_out[t].cpu().numpy()[:len(out)], out.astype(np.uint8)), (seed, "hard bits")
    return dict(seed=seed, C=C, tbs=tbs, qm=qm, nl=nl, G=G, rows=rows, F=cfg.F, n_it=n_it, ok=int(ok.sum()), blocks=int(ok.numel()))


@pytest.mark.parametrize("seed", list(range(10)))
def test_fused_entry_random_geometry_against_the_oracle(dev, seed):
    run_fused_case(seed)


if __name__ == '__main__':
    a, b = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    if len(sys.argv) > 3 and sys.argv[3] == 'fused':
        for sd in range(a, b):
            try:
                print(sd, "ok", run_fused_case(sd), flush=True)
            except Exception as e:
                bad += 1
                print(sd, "FAIL", repr(e)[:400], flush=True)
        print("failed:", bad, flush=True)
        sys.exit(1 if bad else 0)
    for sd in range(a, b):
        try:
            i = run_case(sd)
            print(sd, "ok", {k: i[k] for k in ('numRbs', 'spacing', 'mod', 'layers', 'chan', 'rate', 'bg', 'freqDomain', 'perfect', 'tbs', 'C', 'Zc', 'rows',
                                              'blocks_ok', 'blocks', 'llr_err')}, flush=True)
        except Exception as e:      # (a sweep reports every failing seed)
            bad += 1
            print(sd, "FAIL", draw(sd), repr(e)[:400], flush=True)
    print("failed:", bad, flush=True)
    sys.exit(1 if bad else 0)


"""CPU: the oracle's whole-slot harness (oracle/link.run_slot: Tx chain -> channel -> noise -> OFDM -> LS estimate -> MMSE ->
demap -> rate recovery -> decode -> CRC) pinned AS A WHOLE against the reference-generated slot fixtures
tests/golden/e2e_*.npz (tools/gen_golden.py), on a host without a GPU.

Everything run_slot gets is built on the host: the link tables (DMRS template, RE index, scrambling, pilots, channel static
coefficients, tap matrix) come from neoradium_amd.engine.host_tables -- NumPy code of the class surface, itself pinned
against the reference by tests/test_host_logic.py -- the transport block and the noise from the reference's own random
stream (PCG64 seeded like the fixture: bits -> channel construction -> noise), and the precoder F from the fixture (an SVD
is unique only up to a phase per singular vector).  Checked: LLRs <= 1e-9 of the slot's LLR scale, CRC verdicts, decoded bits."""
import ast
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _build(nr, c, monkeypatch):
    from neoradium_amd import channelmodel
    # the channel objects' per-slot device state (gains/CIR of the current slot) is not needed: host tables only
    monkeypatch.setattr(channelmodel.ChannelModel, 'prepareForNextSlot', lambda self: None)
    nr.random.setSeed(c['seed'])
    car = nr.Carrier(numRbs=c['numRbs'], spacing=c['spacing'])
    bwp = car.curBwp
    p = nr.PDSCH(bwp, numLayers=c['layers'], nID=car.cellId, modulation=c['mod'])
    p.setDMRS(**c['dm'])
    ch_ = c['chan']
    if ch_[0] == 'cdl':
        ch = nr.CdlChannel(bwp, ch_[1], delaySpread=ch_[2], carrierFreq=4e9, dopplerShift=ch_[3],
                           txAntenna=nr.AntennaPanel(ch_[4], polarization='x'), rxAntenna=nr.AntennaPanel(ch_[5], polarization='x'))
    else:
        ch = nr.TdlChannel(bwp, ch_[1], delaySpread=ch_[2], dopplerShift=ch_[3], txAntennaCount=ch_[4], rxAntennaCount=ch_[5])
    return car, bwp, p, ch


@pytest.mark.parametrize("name", ['cfg1_tdl_siso', 'cdl_mimo_td_ls', 'cdl_mimo_fd_perfect', 'cdl_fail_td_ls', 'cfg2_cdl_c_2x2', 'cfg3_cdl_d_4x4_ls',
                                  'cfg3_cdl_d_4x4_perfect'])
def test_oracle_run_slot_vs_reference_slot(name, monkeypatch):
    import neoradium_amd as nr
    from neoradium_amd.engine import host_tables
    from oracle import link as olink
    g = np.load(os.path.join(GOLD, f'e2e_{name}.npz'))
    c = ast.literal_eval(str(g['cfg']))
    car, bwp, p, ch = _build(nr, c, monkeypatch)
    tb_ = host_tables(p, ch, c['rate'], c['bg'])
    assert [w['tbs'] for w in tb_['cw']] == g['tbs'].tolist() and [w['G'] for w in tb_['cw']] == g['G'].tolist()
    assert tb_['max_delay'] == int(g['max_delay'])
    tb = nr.random.bits(tb_['cw'][0]['tbs'])                          # the reference's draw order: bits, then noise
    assert np.array_equal(np.packbits(tb.astype(np.uint8)), g['tb'])
    # the fixture's slot: carrier slot number slot0 (DMRS, scrambling), channel clock at slot min(slot0, 1) -- the reference
    # advances the channel time only once for slot0 goNext() calls in a row (channelmodel.py:180-193, 326, 349)
    slot, chan_slot = c['slot0'], min(c['slot0'], 1)
    st = olink.static_from_tables(tb_, c['numIter'], c['freqDomain'], c['perfect'], True, slots=[slot, chan_slot])
    if c['freqDomain']:
        shape = (tb_['nr'], tb_['L'], tb_['K'])
    else:
        shape = (tb_['nr'], int(tb_['sym_lens'][slot % tb_['slots_per_subframe']][:-1].sum()) + tb_['max_delay'])
    z2 = nr.random.awgn(shape, np.sqrt(2.0))                          # sigma/sqrt(2) = 1: standard-normal complex pairs
    out = olink.run_slot(st, slot, c['snr'], tb.astype(np.int8), z2, F=g['F'], chan_slot=chan_slot)
    assert out['off'] == int(g['t_off'])
    assert abs(out['nv'] - float(g['noise_var'])) <= 1e-9 * float(g['noise_var'])
    ref = g['llr']
    assert np.abs(out['llr'] - ref).max() <= 1e-9 * np.abs(ref).max()
    assert np.array_equal(np.asarray(out['crc'], bool), g['crc'])
    if g['crc'].all():
        assert np.array_equal(np.packbits(np.uint8(out['tb_out'])), g['decoded'])
        assert np.array_equal(out['tb_out'][:-24], tb)

"""Launches on TWO streams at once.  The kernels that hand small device state from launch to launch -- BatchNorm arrival tickets
(csrc/convmodule.hip), the partial sums of the squared-norm reduction (csrc/optim.hip), the barrier counters of the cooperative
LSTM recurrence (csrc/lstm_coop.hip) -- keep it per (device, stream) (csrc/api.hip: emo_stream_scratch); with one area per process
two engines on two streams corrupted each other's tickets / partials.  Every case runs the same work serially on the default
stream and concurrently on two streams (each stream first spins in a sleep kernel while the host queues both streams' launches,
so that the two really overlap on the device) and compares bit for bit."""
from types import SimpleNamespace

import pytest
import torch

from tests.util import CONFIGS, load_golden

pytestmark = pytest.mark.gpu

SPIN = 60_000_000   # cycles of torch.cuda._sleep in front of each stream's work (~30 ms)


def _two_streams(work_a, work_b, rounds):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(SPIN)
    for r in range(rounds):   # alternate, so that both queues fill while the device spins
        with torch.cuda.stream(s1):
            work_a(r)
        with torch.cuda.stream(s2):
            work_b(r)
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    torch.cuda.synchronize()


def test_sqnorm_on_two_streams(dev):
    from emoasr_amd import ops
    torch.manual_seed(0)
    xs = [torch.randn(3_000_001, device=dev), torch.randn(5_000_003, device=dev) * 3.0]
    R = 24
    want = [torch.zeros(R, device=dev), torch.zeros(R, device=dev)]
    for k in range(2):
        for r in range(R):
            ops.sqnorm(xs[k], want[k][r:r + 1])
    torch.cuda.synchronize()
    got = [torch.zeros(R, device=dev), torch.zeros(R, device=dev)]
    _two_streams(lambda r: ops.sqnorm(xs[0], got[0][r:r + 1]), lambda r: ops.sqnorm(xs[1], got[1][r:r + 1]), R)
    for k in range(2):
        assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())
        assert (want[k] - want[k][0]).abs().max().item() == 0.0   # (the reduction is bit-reproducible by itself)


def test_lstm_recurrence_on_two_streams(dev):
    from emoasr_amd import lib, ops
    torch.manual_seed(1)
    dt_ = torch.bfloat16
    shapes = [(9, 36, 512), (7, 20, 256)]
    ins, want, got = [], [], []
    for U, B, H in shapes:
        pre = (torch.randn(U, B, 4 * H, device=dev)).to(dt_)
        w_hh = (torch.randn(4 * H, H, device=dev) * H ** -0.5).to(dt_)
        assert ops.lstm_seq_supported(pre, B, H)
        ins.append((pre, w_hh))
        mk = lambda: (torch.empty(U, B, H, device=dev, dtype=dt_), torch.empty(U, B, H, device=dev),
                      torch.empty(U, B, 4 * H, device=dev, dtype=dt_))
        want.append(mk())
        got.append(mk())
    for k in range(2):
        ops.lstm_seq_fwd(ins[k][0], ins[k][1], None, None, *want[k])
    torch.cuda.synchronize()
    R = 6
    _two_streams(lambda r: ops.lstm_seq_fwd(ins[0][0], ins[0][1], None, None, *got[0]),
                 lambda r: ops.lstm_seq_fwd(ins[1][0], ins[1][1], None, None, *got[1]), R)
    assert lib.size_query("emoasr_lstm_coop_status") == 0
    for k in range(2):
        for a, b in zip(got[k], want[k]):
            assert torch.equal(a, b), k


def test_two_engines_forward_on_two_streams(dev):
    """train-mode forward of two Conformer engines (BatchNorm batch statistics + running-statistics update through the ticket
    kernels, dropout off): encoder outputs and running statistics equal the serial run's, bit for bit"""
    from emoasr_amd.modeling.asr import ASR
    cfg, sd, g = load_golden("l2_tiny")

    def build(seed):
        torch.manual_seed(seed)
        m = ASR(SimpleNamespace(**CONFIGS["l2_tiny"]), compute_dtype=torch.bfloat16)
        m.load_state_dict(sd)
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01 * seed * torch.randn_like(p))
        return m.to(dev).train()

    xs, xlens = g["xs"].to(dev), g["xlens"]
    R = 8

    def run(models, concurrent):
        outs = [[], []]

        def work(k):
            def f(r):
                with torch.no_grad():
                    eouts, _, _ = models[k].encoder(xs, xlens)
                outs[k].append(eouts)
            return f
        if concurrent:
            _two_streams(work(0), work(1), R)
        else:
            for k in range(2):
                for r in range(R):
                    work(k)(r)
            torch.cuda.synchronize()
        return outs

    serial_models = [build(1), build(2)]
    want = run(serial_models, False)
    conc_models = [build(1), build(2)]
    # (first use of a stream's scratch areas allocates: warm every path once outside the overlapped region)
    got = run(conc_models, True)
    for k in range(2):
        for r in range(R):
            assert torch.equal(got[k][r], want[k][r]), (k, r)
        bs, bc = dict(serial_models[k].named_buffers()), dict(conc_models[k].named_buffers())
        for name in bs:
            assert torch.equal(bs[name], bc[name]), (k, name)


def test_f32_and_f32x3_engines_on_two_streams(dev):
    """the f32 product mode is an argument of every call (dtype code EMO_F32X3, round 6), not a library switch: an exact-f32 engine
    and an f32x3 engine working at once on two streams -- forward and backward, launches interleaved call by call -- give what each
    gives alone, bit for bit; and the two modes do differ from each other (the test would also pass if both ran one mode)"""
    from emoasr_amd.modeling.asr import ASR
    cfg, sd, g = load_golden("l2_tiny")

    def build(mode):
        m = ASR(SimpleNamespace(**CONFIGS["l2_tiny"]), compute_dtype=mode)
        m.load_state_dict(sd)
        return m.to(dev).train()

    xs, xlens, ys, ylens = g["xs"].to(dev), g["xlens"], g["ys"], g["ylens"]
    R = 4

    def run(models, concurrent):
        outs = [[], []]

        def work(k):
            def f(r):
                models[k].zero_grad(set_to_none=True)
                loss, _ = models[k](xs, xlens, ys, ylens, None, None)
                loss.backward()
                outs[k].append((loss.detach().clone(), models[k].decoder.output.weight.grad.clone(),
                                models[k].encoder.conv.conv[0].weight.grad.clone()))
            return f
        if concurrent:
            _two_streams(work(0), work(1), R)
        else:
            for k in range(2):
                for r in range(R):
                    work(k)(r)
            torch.cuda.synchronize()
        return outs

    want = run([build(torch.float32), build("f32x3")], False)
    got = run([build(torch.float32), build("f32x3")], True)
    for k in range(2):
        assert torch.equal(got[k][0][0], want[k][0][0]), (k, got[k][0][0].item(), want[k][0][0].item())
        for r in range(R):
            assert torch.equal(got[k][r][0], want[k][r][0]), (k, r)
            for a, b in zip(got[k][r][1:], want[k][r][1:]):
                # (the weight gradients of the split-K products are summed with float atomics: equal to rounding, not bitwise)
                assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item(), (k, r)
    assert not torch.equal(want[0][0][1], want[1][0][1])   # exact f32 and f32x3 are different arithmetic

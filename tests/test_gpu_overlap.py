"""Free-running channel groups (gdg_ctx_set_overlap): the groups' kernels run on streams of their own and are joined lazily.  Results must
not depend on the number of groups, and everything that touches the context after a process call must see its output."""
import numpy as np
import pytest

from helpers import package, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

B, RATE, NCH = 8192, 96000, 7


def make_ctx(pkg, groups):
    ctx = pkg.Context(NCH, B)
    ctx.set_overlap(groups)
    ctx.amps = []
    for c in range(NCH):
        ctx.append_unit(c, "compressor", params=[1, 30, -20])
        ctx.amps.append(ctx.append_unit(c, "power_amp", fir=synth_ir(9000 + 4000 * c, seed=300 + c)))
        ctx.append_unit(c, "reverb", params=[30])
    return ctx


@pytest.mark.parametrize("groups", [2, 3, 7])
def test_groups_do_not_change_the_result(groups):
    pkg = package()
    blocks = 6
    x = np.stack([0.6 * synth_signal(c, blocks * B, RATE) for c in range(NCH)])
    outs = []
    for g in (1, groups):
        ctx = make_ctx(pkg, g)
        d_in, d_out = ctx.alloc(NCH, blocks * B), ctx.alloc(NCH, blocks * B)
        d_in.upload(x)
        d_blk_in, d_blk_out = ctx.alloc(NCH, B), ctx.alloc(NCH, B)
        lr = ctx.alloc(2, B)
        got_lr = []
        for b in range(blocks):
            # strided copies, the chain, and the spatializer reading the chain's output: all ordered through the library's entry points
            ctx._check(pkg.lib().gdg_copy_rows_device(ctx._h, d_blk_in.ptr, B, d_in.ptr + 8 * b * B, blocks * B, B, NCH))
            ctx.process_device(d_blk_in, d_blk_out, B, RATE)
            ctx.spatialize_device(d_blk_out, lr, B)
            ctx._check(pkg.lib().gdg_copy_rows_device(ctx._h, d_out.ptr + 8 * b * B, blocks * B, d_blk_out.ptr, B, B, NCH))
            got_lr.append(lr.download())
        outs.append((d_out.download(), np.concatenate(got_lr, axis=1)))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_parameter_change_between_free_running_calls():
    """A set between two device-resident calls (new plan, new unit state) waits for the groups still running."""
    pkg = package()
    x = np.stack([0.6 * synth_signal(c, 4 * B, RATE) for c in range(NCH)])
    res = []
    for g in (1, 2):
        ctx = make_ctx(pkg, g)
        d_in, d_out = ctx.alloc(NCH, B), ctx.alloc(NCH, B)
        out = []
        for b in range(4):
            d_in.upload(np.ascontiguousarray(x[:, b * B:(b + 1) * B]))
            ctx.process_device(d_in, d_out, B, RATE)
            if b == 1:
                for c in range(NCH):
                    ctx.unit_set_fir(ctx.amps[c], synth_ir(5000 + 100 * c, seed=900 + c))      # power amp: state reset + new spectra
            out.append(d_out.download())
        res.append(np.concatenate(out, axis=1))
        ctx.close()
    assert np.array_equal(res[0], res[1])

"""rt_trace_closest_static (the entry point that allocates nothing) against rt_trace_closest, through the C ABI.
Bit-exact: primitive ids, fan-triangle ids, objects, float t and fp64 t."""
import numpy as np
import pytest

from ray_tracer_amd import capi, scenes, synth

pytestmark = pytest.mark.gpu


def _same(a, b):
    (ha, ta), (hb, tb) = a, b
    return ha.tobytes() == hb.tobytes() and ta.tobytes() == tb.tobytes()


def test_static_staging_matches_the_allocating_entry_point():
    ctx = capi.Context(0)
    mesh, records, cam = scenes.cornell(64, 64)
    ctx.upload_mesh(mesh)
    big = synth.incoherent_rays(300_000, seed=777)
    big[:, 0:3] *= 0.95
    small = synth.incoherent_rays(70_001, seed=4242)
    small[:, 0:3] *= 0.9
    # two calls in a row on the same GPU, the second with fewer rays: the array keeps the first call's contents
    got_big = ctx.trace_closest_static(big, want_t64=True)
    got_small = ctx.trace_closest_static(small, want_t64=True)
    assert _same(got_big, ctx.trace_closest(big, want_t64=True))
    assert _same(got_small, ctx.trace_closest(small, want_t64=True))
    assert (got_big[0]["prim"] != capi.MISS).mean() > 0.5
    # without t64, and the empty batch
    assert ctx.trace_closest_static(small).tobytes() == got_small[0].tobytes()
    assert len(ctx.trace_closest_static(np.zeros((0, 8), np.float32))) == 0
    # a second context on the same GPU and another scene share the array
    other = capi.Context(0)
    other.upload_mesh(synth.soup(20_000, 0.02, seed=3))
    rays = synth.incoherent_rays(capi.STATIC_MAX_RAYS, seed=9)
    assert _same(other.trace_closest_static(rays, want_t64=True), other.trace_closest(rays, want_t64=True))
    assert _same(ctx.trace_closest_static(small, want_t64=True), got_small)
    # beyond the stated maximum: refused, nothing traced
    with pytest.raises(capi.BackendError) as e:
        other.trace_closest_static(np.zeros((capi.STATIC_MAX_RAYS + 1, 8), np.float32))
    assert e.value.code == capi.ERR_INVALID and "RT_STATIC_MAX_RAYS" in str(e.value)
    other.close()
    ctx.close()

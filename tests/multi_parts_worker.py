"""Worker of tests/test_gpu_parity.py::test_multi_device_parts_with_fake_rccl — runs in its OWN process because a process loads one
RCCL: here the fake of tests/fake_rccl (RT_RCCL_LIB) with RT_MULTI_SPLIT_PARTS=1, so that rt_ctx_create_multi builds one PART per list
entry on the one GPU (own context, stream, staging buffer, communicator rank) — the code a real multi-GPU node runs for devices
other than the root.  Prints 'ok' and the fake's call log summary; any mismatch raises."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import capi, scenegen  # noqa: E402
from tests import oracle_py  # noqa: E402


def main():
    fake = C.CDLL(os.environ["RT_RCCL_LIB"], mode=C.RTLD_GLOBAL)
    fake.fake_rccl_log.restype = C.c_size_t
    fake.fake_rccl_log.argtypes = [C.c_char_p, C.c_size_t]

    def take_log():
        n = fake.fake_rccl_log(None, 0)
        buf = C.create_string_buffer(n + 1)
        fake.fake_rccl_log(buf, n + 1)
        fake.fake_rccl_log_reset()
        return buf.value.decode().splitlines()

    import torch
    oracle = oracle_py.load()
    w, h = 640, 361  # odd height: uneven bands
    objs, lights = scenegen.generate(w, h, 1300, 4, seed=21)  # two LDS chunks
    m = oracle.default_camera(w, h)
    with capi.Context(0) as one:
        one.scene_upload(objs, lights)
        ref, ref_hit = one.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
        ref_gen = one.render(m, w, h, 2, capi.RT_SHADOWS)
    assert np.array_equal(ref, oracle.render(objs, lights, m, w, h))
    for parts in (2, 4, 8):
        take_log()
        with capi.Context(device_ids=[0] * parts) as ctx:
            assert take_log() == ["init %d:%s" % (parts, " 0" * parts)], "one communicator rank per part"
            ctx.scene_upload(objs, lights)
            got, hit = ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
            assert np.array_equal(got, ref), f"{parts} parts: image differs"
            assert np.array_equal(hit, ref_hit), f"{parts} parts: hit plane differs"
            log = take_log()
            # one group; per band g > 0 a receive on rank 0 from rank g and a send of rank g to rank 0, pixels then hit indices
            assert log[0] == "group" and log[-1] == "end 0" and log.count("group") == 1
            want = []
            for g in range(1, parts):
                rows = (h * (g + 1)) // parts - (h * g) // parts
                for nbytes in (rows * w * 4, rows * w * 4):  # RGBA8 pixels, int32 hit indices
                    want += ["recv 0<-%d %d" % (g, nbytes), "send %d->0 %d" % (g, nbytes)]
            assert log[1:-1] == want, (log, want)
            # general path, RGB8, a short band (fewer rows than parts), and frames back to back on a caller stream
            assert np.array_equal(ctx.render(m, w, h, 2, capi.RT_SHADOWS), ref_gen), f"{parts} parts: general path"
            assert np.array_equal(ctx.render(m, w, h, 1, capi.RT_OUT_RGB8), ref[..., :3])
            assert np.array_equal(ctx.render_rows(m, w, h, 100, 103)[0], ref[100:103])
            take_log()
            st = torch.cuda.Stream()
            frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
            for k in range(12):  # every part's staging is reused frame after frame: the send of frame k must be over before render k + 1
                ctx.render_rows_device(m, w, h, 0, h, 1, 0, frames[k % 4].data_ptr(), 0, st.cuda_stream)
            st.synchronize()
            for f in frames:
                assert np.array_equal(f.cpu().numpy(), ref), f"{parts} parts: pipelined frames"
            assert take_log().count("end 0") == 12
    print("ok")


if __name__ == "__main__":
    main()

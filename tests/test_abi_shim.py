"""Host-only behaviour of the libimageflow C-ABI subset (csrc/abi_shim.cpp): version handshake, io table, JSON reader,
sticky errors with the reference's category / HTTP / exit codes (imageflow_core/src/errors.rs:779-902), response
envelopes (json/mod.rs:158-181).  No GPU: jobs that need the device must fail loudly, not fall back."""
import ctypes as C

import pytest

from imageflow_amd import abi
from imageflow_amd.abi import Context


def test_version_handshake():
    L = abi._bind()
    assert L.imageflow_abi_version_major() == 3 and L.imageflow_abi_version_minor() == 2     # abi_version.rs:4,7
    assert L.imageflow_abi_compatible(3, 2) and L.imageflow_abi_compatible(3, 0)
    assert not L.imageflow_abi_compatible(2, 0) and not L.imageflow_abi_compatible(3, 9)
    assert not L.imageflow_context_create(2, 0)


def test_get_version_info_and_unknown_endpoint():
    with Context() as c:
        status, r = c.send_json("v1/get_version_info", {})
        assert status == 200 and r["success"] is True and "gfx950" in r["data"]["version_info"]["long_version_string"]
        assert not c.has_error()
        status, r = c.send_json("v1/teapot", {})
        assert status == 404 and r["message"] == "Endpoint name not understood"
        assert c.has_error()


@pytest.mark.parametrize("body", [b"{bad", b"", b'{"framewise": ', b'{"a": 1} trailing', b'{"framewise":{"steps":[{"decode":{"io_id":"x"}}]}}',
                                  b"[" * 100 + b"]" * 100])
def test_invalid_json_is_category_3_http_400_exit_65(body):
    with Context() as c:
        status, r = c.send_json("v1/execute", body)
        assert status == 400 and r["success"] is False and r["message"].startswith("InvalidJson")
        assert c.has_error() and c.error_code() == 3
        assert c.L.imageflow_context_error_as_http_code(c.p) == 400 and c.L.imageflow_context_error_as_exit_code(c.p) == 65
        assert c.L.imageflow_context_error_recoverable(c.p) and c.L.imageflow_context_error_try_clear(c.p) and not c.has_error()


def test_first_error_sticks_and_truncated_message():
    with Context() as c:
        c.send_json("v1/execute", b"{bad")
        first, whole = c.error_message()
        c.send_json("v1/nope", {})
        assert c.error_message()[0] == first and whole and c.error_code() == 3
        b = C.create_string_buffer(20)
        n = C.c_size_t()
        assert not c.L.imageflow_context_error_write_to_buffer(c.p, b, 20, C.byref(n))
        assert b.value.endswith(b"\n[truncated]\n") and n.value == len(b.value) == 19


def test_io_table_rules():
    with Context() as c:
        assert c.add_input_buffer(0, b"\xff\xd8\xff" + bytes(10))
        assert not c.add_input_buffer(0, b"abc") and c.has_error() and c.error_code() == 2
    with Context() as c:
        assert c.add_output_buffer(1) and not c.add_output_buffer(1)
    with Context() as c:
        assert c.add_output_buffer(1)
        assert c.get_output_buffer(1) == b""                                   # nothing written yet
        assert c.get_output_buffer(7) is None and c.has_error()
    with Context() as c:                                                        # placeholder without a registered buffer
        status, r = c.send_json("v1/build", {"io": [{"io_id": 0, "direction": "in", "io": "placeholder"}],
                                             "framewise": {"steps": []}})
        assert status == 400 and "placeholder" in r["message"]


def test_nodes_outside_the_hot_path_answer_action_not_supported():
    with Context() as c:
        c.add_input_buffer(0, b"GIF89a" + bytes(32))
        c.add_output_buffer(1)
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        assert status == 400 and c.error_code() == 5 and "ImageTypeNotSupported" in r["message"]
    with Context() as c:
        status, r = c.send_json("v1/execute", {"framewise": {"graph": {"nodes": {"0": {"white_balance_histogram_area_threshold_srgb": {}}}, "edges": []}}})
        assert status == 400 and c.error_code() in (7, 8)


def test_no_cpu_fallback_for_jobs():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with Context() as c:
        c.add_output_buffer(1)
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [
            {"create_canvas": {"w": 8, "h": 8, "format": "bgra_32", "color": "transparent"}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        assert status == 500 and r["success"] is False and c.has_error()
        assert c.get_output_buffer(1) == b""


def test_context_memory():
    with Context() as c:
        p = c.L.imageflow_context_memory_allocate(c.p, 100, None, 0)
        assert p and c.L.imageflow_context_memory_free(c.p, p, None, 0) and not c.L.imageflow_context_memory_free(c.p, p, None, 0)


def test_json_reader_survives_mutated_jobs():
    """Untrusted bytes in `send_json`: byte mutations, truncations and deep nesting of real job bodies answer with an error
    envelope (or get as far as needing the device) -- the process stays up and the context stays usable."""
    import json
    import random
    rnd = random.Random(11)
    jobs = [
        {"io": [{"io_id": 0, "direction": "in", "io": "placeholder"}, {"io_id": 1, "direction": "out", "io": "output_buffer"}],
         "framewise": {"steps": [{"decode": {"io_id": 0, "commands": [{"jpeg_downscale_hints": {"width": 800, "height": 600, "scale_luma_spatially": True,
                                                                                                "gamma_correct_for_srgb_during_spatial_luma_scaling": True}}]}},
                                 {"resample_2d": {"w": 200, "h": 200, "hints": {"down_filter": "robidoux", "scaling_colorspace": "linear", "sharpen_percent": 15,
                                                                                  "background_color": {"srgb": {"hex": "FFFFFFFF"}}}}},
                                 {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": {"quality": 90, "progressive": True}}}}]}},
        {"framewise": {"graph": {"nodes": {"0": {"create_canvas": {"w": 64, "h": 64, "format": "bgra_32", "color": "transparent"}},
                                           "1": {"fill_rect": {"x1": 0, "y1": 0, "x2": 10, "y2": 10, "color": {"srgb": {"hex": "EECCFFFF"}}}},
                                           "2": {"constrain": {"mode": "within", "w": 32}}, "3": {"command_string": {"kind": "ir4", "value": "width=20&mode=max"}}},
                                 "edges": [{"from": 0, "to": 1, "kind": "input"}, {"from": 1, "to": 2, "kind": "input"}, {"from": 2, "to": 3, "kind": "input"}]}}},
    ]
    answered = 0
    for job in jobs:
        body = json.dumps(job).encode()
        with Context() as c:
            c.add_input_buffer(0, b"\xff\xd8\xff" + bytes(16))
            c.add_output_buffer(1)
            for _ in range(600):
                m = bytearray(body)
                for _ in range(rnd.randint(1, 5)):
                    m[rnd.randrange(len(m))] = rnd.choice(b'{}[]",:0123456789truefalsn\\ \x00\xff' + bytes([rnd.randrange(256)]))
                if rnd.random() < 0.2:
                    m = m[:rnd.randrange(1, len(m))]
                status, r = c.send_json(rnd.choice(["v1/execute", "v1/build"]), bytes(m))
                assert status in (200, 400, 404, 500, 501, 503), status
                assert r is None or "success" in r
                answered += 1
                c.L.imageflow_context_error_try_clear(c.p)
            status, r = c.send_json("v1/execute", b"[" * 5000)          # nesting far beyond any job
            assert status == 400
            assert c.send_json("v1/get_version_info", {})[0] in (200, 400)      # still answering
    assert answered == 1200

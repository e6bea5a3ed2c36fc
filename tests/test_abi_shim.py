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
        # JsonResponse::fail_with_message serialises ResponsePayload::None, a unit variant renamed "none" (json/mod.rs:169-180,
        # imageflow_types/src/lib.rs:2061-2062): the reference's Response001 deserialises "none", not {}
        assert r["code"] == 400 and r["data"] == "none"
        assert c.has_error() and c.error_code() == 3
        assert c.L.imageflow_context_error_as_http_code(c.p) == 400 and c.L.imageflow_context_error_as_exit_code(c.p) == 65
        # FlowError::recoverable() is `false` for every error (imageflow_core/src/errors.rs:656-658, 953-969): an error,
        # once set, stays -- try_clear only answers true while there is nothing to clear
        assert not c.L.imageflow_context_error_recoverable(c.p) and not c.L.imageflow_context_error_try_clear(c.p) and c.has_error()
    with Context() as c:
        assert c.L.imageflow_context_error_recoverable(c.p) and c.L.imageflow_context_error_try_clear(c.p)


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
            status, r = c.send_json("v1/execute", b"[" * 5000)          # nesting far beyond any job
            assert status == 400
            assert c.send_json("v1/get_version_info", {})[0] in (200, 400)      # still answering
    assert answered == 1200


def test_header_matches_the_reference_header_symbol_for_symbol():
    """include/imageflow_abi_subset.h declares, and the library exports, every function of the reference's generated
    header (bindings/headers/imageflow_default.h: 25 functions; the list is restated here because /root/reference does
    not travel to the GPU box)."""
    import os
    import re
    reference_functions = """imageflow_abi_compatible imageflow_abi_version_major imageflow_abi_version_minor imageflow_buffer_free
        imageflow_context_add_input_buffer imageflow_context_add_output_buffer imageflow_context_begin_terminate imageflow_context_create
        imageflow_context_destroy imageflow_context_error_as_exit_code imageflow_context_error_as_http_code imageflow_context_error_code
        imageflow_context_error_recoverable imageflow_context_error_try_clear imageflow_context_error_write_to_buffer
        imageflow_context_get_output_buffer_by_id imageflow_context_has_error imageflow_context_memory_allocate imageflow_context_memory_free
        imageflow_context_print_and_exit_if_error imageflow_context_request_cancellation imageflow_context_send_json
        imageflow_context_take_output_buffer imageflow_json_response_destroy imageflow_json_response_read""".split()
    assert len(reference_functions) == 25
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ours = set(re.findall(r"\b(imageflow_[a-z_0-9]+)\s*\(", open(os.path.join(root, "include", "imageflow_abi_subset.h")).read()))
    assert set(reference_functions) <= ours, sorted(set(reference_functions) - ours)
    ref_header = "/root/reference/bindings/headers/imageflow_default.h"
    if os.path.exists(ref_header):                                     # in the build container: the restated list is the header's
        txt = re.sub(r"//[^\n]*", "", open(ref_header).read())
        theirs = set(re.findall(r"\b(imageflow_[a-z_0-9]+)\s*\(", txt))
        assert theirs == set(reference_functions), sorted(theirs ^ set(reference_functions))
    L = abi._bind()
    for name in reference_functions:
        assert hasattr(L, name), name


def test_take_output_buffer_state_machine():
    """CodecInstanceContainer's Ready / Lent / Taken (imageflow_core/src/codecs/mod.rs:421-436,560-626)."""
    with Context() as c:
        assert c.add_output_buffer(1)
        assert c.take_output_buffer(1) == b""                                  # Ready -> Taken (empty: nothing was encoded)
        assert c.take_output_buffer(1) is None and c.error_code() == 2         # already taken
        assert "already been taken" in c.error_message()[0]
    with Context() as c:
        assert c.add_output_buffer(1)
        assert c.get_output_buffer(1) == b"" and c.get_output_buffer(1) == b""   # Lent, idempotent
        assert c.take_output_buffer(1) is None and "lent out" in c.error_message()[0]
    with Context() as c:
        assert c.add_output_buffer(1)
        assert c.take_output_buffer(1) == b""
        assert c.get_output_buffer(1) is None and "already been taken" in c.error_message()[0]
    with Context() as c:
        c.add_input_buffer(0, b"\xff\xd8\xff")
        assert c.take_output_buffer(0) is None and c.take_output_buffer(9) is None and c.error_code() == 2
        assert not c.L.imageflow_context_take_output_buffer(c.p, 0, None, None)
    assert abi._bind().imageflow_buffer_free(None, 0)                           # NULL is a no-op, always true


def test_job_with_cancellation():
    """imageflow_abi/src/lib.rs:1628-1665 (test_job_with_cancellation): cancel, then send -- a response comes back and
    the context carries category 21 (HTTP 499, exit 130; errors.rs:836,873,901)."""
    with Context() as c:
        c.add_input_buffer(0, b"\xff\xd8\xff" + bytes(16))
        c.add_output_buffer(1)
        c.request_cancellation()
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [
            {"decode": {"io_id": 0}}, "flip_h", "rotate_90", {"resample_2d": {"w": 30, "h": 20, "hints": {"sharpen_percent": None}}},
            {"constrain": {"mode": "within", "w": 5, "h": 5}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        assert status == 499 and r["success"] is False and r["message"].startswith("OperationCancelled")
        assert c.error_code() == 21
        assert c.L.imageflow_context_error_as_http_code(c.p) == 499 and c.L.imageflow_context_error_as_exit_code(c.p) == 130
        assert c.send_json("v1/execute", {"framewise": {"steps": []}})[0] == 499     # nothing runs on a cancelled context


def test_print_and_exit_if_error():
    """lib.rs:744: false without an error; with one, the message goes to stderr and the process exits with the exit code."""
    import subprocess
    import sys
    with Context() as c:
        assert c.L.imageflow_context_print_and_exit_if_error(c.p) is False
    code = ("from imageflow_amd.abi import Context\n"
            "c = Context()\n"
            "c.send_json('v1/execute', b'{bad')\n"
            "c.L.imageflow_context_print_and_exit_if_error(c.p)\n"
            "print('still here')\n")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root)
    assert r.returncode == 65 and "InvalidJson" in r.stderr and "still here" not in r.stdout


@pytest.mark.parametrize("number", [b"inf", b"Infinity", b"nan", b"0x10", b"+1", b"1e999", b"01", b"1.", b".5", b"-"])
def test_json_numbers_follow_the_grammar(number):
    with Context() as c:
        status, r = c.send_json("v1/execute", b'{"framewise":{"steps":[{"create_canvas":{"w":' + number + b',"h":8,"format":"bgra_32","color":"transparent"}}]}}')
        assert status == 400 and c.error_code() == 3, (number, r)


def test_frame_size_limits_are_the_reference_defaults():
    """ExecutionSecurity::sane_defaults (imageflow_types/src/lib.rs:1226-1236): frames above 10000 px a side or 100 MP are
    refused before anything is allocated; SizeLimitExceeded is category 2 (errors.rs:217)."""
    def canvas(w, h, **extra):
        with Context() as c:
            msg = {"framewise": {"steps": [{"create_canvas": {"w": w, "h": h, "format": "bgra_32", "color": "transparent"}}]}}
            msg.update(extra)
            status, r = c.send_json("v1/execute", msg)
            return status, r["message"], c.error_code()
    for w, h in ((10001, 8), (8, 10001), (2 ** 31 - 1, 1), (2 ** 30, 2 ** 30)):
        status, msg, code = canvas(w, h)
        assert status == 400 and code == 2 and msg.startswith("SizeLimitExceeded"), (w, h, msg)
    status, msg, code = canvas(10000, 10000)[0:3]
    assert status != 200 and (msg.startswith("SizeLimitExceeded") or "Gpu" in msg or "GPU" in msg or "hip" in msg.lower())   # 100 MP exactly passes the limit
    status, msg, code = canvas(64, 64, security={"max_frame_size": {"w": 32, "h": 32, "megapixels": 1}})
    assert status == 400 and "max_frame_size.w 32" in msg
    with Context() as c:                                                        # expand_canvas sums cannot wrap
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [
            {"create_canvas": {"w": 2 ** 31 - 1, "h": 4, "format": "bgra_32", "color": "transparent"}}]}})
        assert status == 400
    assert abi._bind().ifhip_stride_for_width(2 ** 30) == 0 and abi._bind().ifhip_stride_for_width(200) == 832


def test_tell_decoder_and_get_scaled_image_info():
    """json/endpoints/v1.rs:347-371: v1/tell_decoder stores the jpeg_downscale_hints with the input's decoder,
    v1/get_scaled_image_info answers with the size MzDec::apply_downscaling (mozjpeg_decoder.rs:588-618) would decode at --
    header parsing only, no GPU."""
    import numpy as np
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_entropy_cases.npz"))
    data = z["jpg_0"].tobytes()
    with Context() as c:
        c.add_input_buffer(0, data)
        status, r = c.send_json("v1/get_image_info", {"io_id": 0})
        assert status == 200
        w, h = r["data"]["image_info"]["image_width"], r["data"]["image_info"]["image_height"]
        status, r = c.send_json("v1/get_scaled_image_info", {"io_id": 0})                  # nothing told yet: the full size
        assert status == 200 and (r["data"]["image_info"]["image_width"], r["data"]["image_info"]["image_height"]) == (w, h)
        hints = {"width": max(1, w // 3), "height": max(1, h // 3), "scale_luma_spatially": True,
                 "gamma_correct_for_srgb_during_spatial_luma_scaling": True}
        status, r = c.send_json("v1/tell_decoder", {"io_id": 0, "command": {"jpeg_downscale_hints": hints}})
        assert status == 200 and r["success"] is True and r["data"] == {}
        status, r = c.send_json("v1/get_scaled_image_info", {"io_id": 0})
        sw, sh = r["data"]["image_info"]["image_width"], r["data"]["image_info"]["image_height"]
        exp = next(((-(-w * i // 8), -(-h * i // 8)) for i in (1, 2, 3, 4, 5, 6) if -(-w * i // 8) >= hints["width"] and -(-h * i // 8) >= hints["height"]),
                   (w, h))
        assert status == 200 and (sw, sh) == exp and (sw, sh) != (w, h)
        status, r = c.send_json("v0.1/tell_decoder", {"io_id": 0, "command": "discard_color_profile"})     # accepted, nothing to act on
        assert status == 200 and not c.has_error()
        status, r = c.send_json("v1/tell_decoder", {"io_id": 0, "command": {"make_coffee": {}}})
        assert status == 400 and c.error_code() == 3
    with Context() as c:
        status, r = c.send_json("v1/tell_decoder", {"io_id": 5, "command": "discard_color_profile"})        # no such input
        assert status == 400 and c.has_error()


def test_byte_array_io_is_validated():
    with Context() as c:
        status, r = c.send_json("v1/build", {"io": [{"io_id": 0, "direction": "in", "io": {"byte_array": [1, 2, 300]}}], "framewise": {"steps": []}})
        assert status == 400 and c.error_code() == 3                                       # InvalidJson
    with Context() as c:
        status, r = c.send_json("v1/build", {"io": [{"io_id": 0, "direction": "in", "io": "output_base_64"}], "framewise": {"steps": []}})
        assert status == 400 and c.error_code() == 3


def test_contexts_and_devices_without_a_gpu():
    """ifhip_shim_spread_contexts / _context_set_device (include/imageflow_abi_subset.h): with no usable device a context
    stays on "the calling thread's device" (-1) whatever the policy says, and binding it to an ordinal is an InvalidArgument
    on that context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form is tests/test_gpu_abi_shim.py::test_contexts_bound_to_devices")
    L = abi._bind()
    L.ifhip_shim_spread_contexts(1)
    try:
        with Context() as c:
            assert c.device == -1
            assert not c.set_device(0)
            assert c.has_error() and "device ordinal 0" in c.error_message()[0]
        with Context() as c:
            assert c.set_device(-1) and not c.has_error()
    finally:
        L.ifhip_shim_spread_contexts(0)

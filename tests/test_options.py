"""CPU: ``ops.options`` -- the one supported way to flip the package's process-wide switches (conv precision / algorithm, weight-gradient
algorithm, fetch lists, dispatch log): values restored on exit, also when the body raises; unknown keys / values rejected up front;
re-entrant; a second thread entering a block waits for the first."""
import threading
import time

import pytest


def test_options_restore_on_exit_and_on_error():
    from deep_fluids_amd import ops
    assert (ops.CONV_PRECISION, ops.CONV_ALGO, ops.WGRAD_ALGO, ops.ACTIVATION_FETCH) == ("fp32", "auto", 0, None)
    fetch = []
    with ops.options(conv_precision="bf16x3", wgrad_algo=4, activation_fetch=fetch):
        assert ops.CONV_PRECISION == "bf16x3" and ops.WGRAD_ALGO == 4 and ops.ACTIVATION_FETCH is fetch
        with ops.options(conv_algo="direct", wgrad_algo=1):           # nested, same thread: re-entrant
            assert (ops.CONV_ALGO, ops.WGRAD_ALGO, ops.CONV_PRECISION) == ("direct", 1, "bf16x3")
        assert (ops.CONV_ALGO, ops.WGRAD_ALGO) == ("auto", 4)
    assert (ops.CONV_PRECISION, ops.CONV_ALGO, ops.WGRAD_ALGO, ops.ACTIVATION_FETCH) == ("fp32", "auto", 0, None)
    with pytest.raises(RuntimeError):
        with ops.options(conv_algo="winograd", dispatch_counts={}):
            raise RuntimeError("a failing test body")
    assert ops.CONV_ALGO == "auto" and ops.DISPATCH_COUNTS is None


def test_options_reject_unknown_keys_and_values_before_changing_anything():
    from deep_fluids_amd import ops
    with pytest.raises(TypeError):
        with ops.options(conv_algo="direct", conv_precison="bf16x3"):
            pass
    assert ops.CONV_ALGO == "auto"
    with pytest.raises(ValueError):
        with ops.options(wgrad_algo=1, conv_precision="bf16"):
            pass
    assert ops.WGRAD_ALGO == 0 and ops.CONV_PRECISION == "fp32"


def test_options_blocks_serialise_across_threads():
    from deep_fluids_amd import ops
    seen = []
    inside = threading.Event()

    def other():
        inside.wait()
        with ops.options(conv_algo="direct"):
            seen.append(("other", ops.CONV_ALGO, ops.WGRAD_ALGO))

    t = threading.Thread(target=other)
    t.start()
    with ops.options(wgrad_algo=2):
        inside.set()
        time.sleep(0.1)                      # the other thread is waiting on the lock, not interleaving
        seen.append(("main", ops.CONV_ALGO, ops.WGRAD_ALGO))
    t.join()
    assert seen == [("main", "auto", 2), ("other", "direct", 0)]
    assert (ops.CONV_ALGO, ops.WGRAD_ALGO) == ("auto", 0)

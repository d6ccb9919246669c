"""The render kernels' reciprocal, square root and 1 / sqrt (dev_vec.hpp: rcp_cr, sqrt_cr, rsqrt_cr -- the compiler's correctly
rounded sequences without their operand guards when every active lane's operand lies in [2^-60, 2^60], the compiler's sequences
otherwise) against `1.0f / x`, `sqrtf(x)`, `1.0f / sqrtf(x)` as hipcc compiles them: EVERY one of the 2^32 float bit patterns,
on the hardware, bit for bit.  Unary functions can be checked exhaustively; that is why only these took the short cut.  Ops 3 .. 6: a quotient by a divisor known when the
code is written (dev_vec.hpp: DivConst -- product with RN(1 / c) and one residual correction), one op per (c, y) pair in use."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op,name", [(0, "1/x"), (1, "sqrt"), (2, "1/sqrt"), (3, "x/pi"), (4, "x/0.01^2"), (5, "x/0.02^2"), (6, "x/0.1^2")])
def test_every_float(gpu_hooks, op, name):
    bad, first = gpu_hooks.unary_test(op, 0, 1 << 32)
    assert bad == 0, f"{name}: {bad} operands differ, the smallest has bit pattern {first:#010x}"


def test_lanes_of_one_wavefront_in_and_out_of_range(gpu_hooks):
    """consecutive bit patterns put a whole wavefront on one side of the range test; a stride that mixes exponents inside a
    wavefront is covered by the range's edges: 64 patterns around each edge of [2^-60, 2^60], both signs"""
    for op in (0, 1, 2, 3, 4, 5, 6):
        for edge in (0x21800000, 0x5D800000, 0xA1800000, 0xDD800000, 0x00800000, 0x7F800000, 0x0F800000):
            bad, first = gpu_hooks.unary_test(op, edge - 96, 192)
            assert bad == 0, (op, hex(edge), hex(first))

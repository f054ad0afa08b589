"""Host-side pieces of the fused pass that need no GPU: expression trees -> hy_expression (postfix), their result types, and the
2-rank merge's bookkeeping of which partial feeds which aggregate."""
import numpy as np
import pytest

from hyrise_amd import abi
from hyrise_amd.distributed import expression_type
from hyrise_amd.operators import expression


class Column:   # what operators.expression needs of a device column
    def __init__(self, handle, data_type):
        self.handle, self.data_type = handle, data_type


PRICE, DISCOUNT, TAX, KEY = Column(0x1000, abi.TYPE_FLOAT), Column(0x2000, abi.TYPE_FLOAT), Column(0x3000, abi.TYPE_FLOAT), Column(0x4000, abi.TYPE_LONG)
ONE = (abi.TYPE_INT, 1)
DISC_PRICE = (abi.ARITH_MUL, PRICE, (abi.ARITH_SUB, ONE, DISCOUNT))
CHARGE = (abi.ARITH_MUL, DISC_PRICE, (abi.ARITH_ADD, ONE, TAX))


def nodes_of(e):
    out = []
    for n in e.nodes[:e.n_nodes]:
        if n.kind == abi.EXPR_COLUMN:
            out.append(("column", n.column))
        elif n.kind == abi.EXPR_LITERAL:
            out.append(("literal", n.literal_type, n.literal.i32 if n.literal_type == abi.TYPE_INT else None))
        else:
            out.append(("op", n.op))
    return out


def test_expression_trees_become_postfix_programs():
    """tpch_queries.cpp:60-80: l_extendedprice * (1 - l_discount) and ... * (1 + l_tax) -- the operands of an operator come before it,
    left before right (the kernel's stack computes slot 1 <op> slot 0)."""
    assert nodes_of(expression(PRICE)) == [("column", 0x1000)]
    assert nodes_of(expression(DISC_PRICE)) == [("column", 0x1000), ("literal", abi.TYPE_INT, 1), ("column", 0x2000), ("op", abi.ARITH_SUB), ("op", abi.ARITH_MUL)]
    charge = nodes_of(expression(CHARGE))
    assert charge == [("column", 0x1000), ("literal", abi.TYPE_INT, 1), ("column", 0x2000), ("op", abi.ARITH_SUB), ("op", abi.ARITH_MUL),
                      ("literal", abi.TYPE_INT, 1), ("column", 0x3000), ("op", abi.ARITH_ADD), ("op", abi.ARITH_MUL)]
    assert len(charge) <= abi.MAX_EXPRESSION_NODES
    null_literal = expression((abi.ARITH_ADD, PRICE, None))
    assert null_literal.nodes[1].kind == abi.EXPR_LITERAL and null_literal.nodes[1].literal_type == abi.TYPE_NULL
    too_long = PRICE
    for _ in range(7):
        too_long = (abi.ARITH_ADD, too_long, ONE)   # 15 nodes
    with pytest.raises(ValueError):
        expression(too_long)


def test_expression_types_follow_expression_common_type():
    """expression_utils.cpp:172-204: double wins, long with float gives double, NULL takes the other side's type."""
    assert expression_type(DISC_PRICE) == abi.TYPE_FLOAT and expression_type(CHARGE) == abi.TYPE_FLOAT
    assert expression_type((abi.ARITH_ADD, KEY, (abi.TYPE_INT, 7))) == abi.TYPE_LONG
    assert expression_type((abi.ARITH_MUL, KEY, PRICE)) == abi.TYPE_DOUBLE
    assert expression_type((abi.ARITH_MUL, PRICE, (abi.TYPE_DOUBLE, 0.5))) == abi.TYPE_DOUBLE
    assert expression_type((abi.ARITH_ADD, PRICE, None)) == abi.TYPE_FLOAT
    assert expression_type((abi.ARITH_SUB, (abi.TYPE_INT, 1), (abi.TYPE_INT, 2))) == abi.TYPE_INT
    # ... the same rule the oracle states
    from support import oracle
    for left in (abi.TYPE_NULL, abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE):
        for right in (abi.TYPE_NULL, abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE):
            if left == right == abi.TYPE_NULL:
                continue
            tree = (abi.ARITH_ADD, None if left == abi.TYPE_NULL else (left, 1), None if right == abi.TYPE_NULL else (right, 1))
            assert expression_type(tree) == oracle().hyo_expression_common_type(left, right), (left, right)

"""Shape tables shared by the tests (state_dict key -> shape)."""


def mlp_shapes(d, pre=None, hidden=128):
    """MLP state_dict shapes (NN.py:98-106)."""
    i = d + 1 + (1 if pre else 0)
    return {"main.0.weight": (hidden, i), "main.0.bias": (hidden,), "main.2.weight": (hidden, hidden),
            "main.2.bias": (hidden,), "main.4.weight": (hidden, hidden), "main.4.bias": (hidden,),
            "main.6.weight": (d, hidden), "main.6.bias": (d,)}

def maybe_allow_in_graph(cls):
    return cls

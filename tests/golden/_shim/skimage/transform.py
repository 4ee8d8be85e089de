def rescale(*a, **k):
    raise NotImplementedError("stand-in")

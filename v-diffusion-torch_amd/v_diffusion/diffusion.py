class GaussianDiffusion:  # placeholder, replaced below in this round
    pass


def get_logsnr_schedule(*a, **k):
    raise NotImplementedError

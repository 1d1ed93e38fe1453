class UNet:  # placeholder
    pass

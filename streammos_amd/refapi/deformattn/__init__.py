__smos_refapi__ = True

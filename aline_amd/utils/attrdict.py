class AttrDict(dict):
    """dict with attribute access -- the batch / output container of the reference
    (third-party `attrdictionary.AttrDict` there; model/base.py:3)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self

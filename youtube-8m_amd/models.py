"""Plugin base class (mirrors W/models.py:18-21)."""


class BaseModel(object):
    """Every model plugin derives from this and overrides create_model().

    Contract (W/train.py:361-377, W/eval.py:168-182): called as
    ``create_model(model_input, num_frames=..., vocab_size=..., labels=..., **more)``; unknown keyword
    arguments must be swallowed; the result is a dict with at least ``"predictions"`` ([B, V] probabilities).
    """

    def create_model(self, unused_model_input, **unused_params):
        raise NotImplementedError()

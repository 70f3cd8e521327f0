"""Plain (NumPy-era) target interface, kept so legacy targets can still be type-checked.

Mirrors interfaces/target.py:1-14 of the reference: a `density(x)` / `draw_sample(beta)` pair.
The HIP engine cannot evaluate such targets; the GPU samplers reject them loudly.
"""


class TargetDistribution:
    def __init__(self, dimension):
        self.dim = dimension

    def density(self, x):
        raise NotImplementedError("Subclasses must implement the density method.")

    def draw_sample(self, beta=1.0):
        raise NotImplementedError("Subclasses must implement the draw_sample method.")

"""The reference's validity / MatterSim / SUN filters wrap external evaluators that need network assets
(SURVEY.md section 2 row 13): out of scope.  A pass-through keeps configs that name a filter loadable."""


def invalid_filter(sample_data, sample_struc=None):
    return sample_data, sample_struc


class OptFilter:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, sample_data, sample_struc, energies=None):
        return sample_data, sample_struc, {}

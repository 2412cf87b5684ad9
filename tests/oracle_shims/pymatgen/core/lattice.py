class Lattice:  # placeholder; never constructed by the golden generator
    pass

class Structure:  # placeholder; never constructed by the golden generator
    pass

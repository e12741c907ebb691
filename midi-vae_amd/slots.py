"""Slots of the engine's scalar accumulator block (losses / hit counts of a step) and host-side input modes."""
# slots of the scalar accumulator
S_NOTES_LOSS, S_NOTES_HITS, S_INSTR_LOSS, S_INSTR_HITS, S_VEL_LOSS, S_VEL_HITS, S_KL, S_STYLE_LOSS, S_STYLE_HITS = range(9)
S_HELD_LOSS, S_HELD_HITS, S_NEXT_LOSS, S_NEXT_HITS = 10, 11, 12, 13
S_SIG_LOSS, S_SIG_HITS, S_CNOTES_LOSS, S_CNOTES_HITS, S_CINSTR_LOSS, S_CINSTR_HITS = 14, 15, 16, 17, 18, 19
N_SCALARS = 32
X_EXT = 100     # (host-side only) a recurrent layer whose x*W + b is written by the caller: classifiers on the decoder's OUTPUTS
X_GATHER2 = 101  # (host-side only) two-hot input rows: two table rows per step, summed into x*W + b before the recurrence

"""Mirror of the reference's `msbwt_core` module (src/msbwt_core.rs): alphabet constants,
`BWTRange`, and the `BWT` interface the GPU index implements."""
from dataclasses import dataclass

VC_LEN = 6        # $ A C G N T           (src/msbwt_core.rs:4)
LETTER_BITS = 3   #                        (:6)
NUMBER_BITS = 5   #                        (:8)
NUM_POWER = 32    #                        (:10)
MASK = 0x07       #                        (:12)
COUNT_MASK = 0x1F  #                       (:14)


@dataclass(frozen=True)
class BWTRange:
    """Half-open range [l, h) of the BWT (src/msbwt_core.rs:18-24)."""
    l: int = 0
    h: int = 0


class BWT:
    """The trait surface of src/msbwt_core.rs:28-162."""

    def load_vector(self, bwt):
        raise NotImplementedError

    def load_numpy_file(self, filename):
        raise NotImplementedError

    def get_symbol_count(self, symbol):
        raise NotImplementedError

    def get_total_size(self):
        raise NotImplementedError

    def constrain_range(self, sym, input_range):
        raise NotImplementedError

    def count_kmer(self, kmer):
        raise NotImplementedError

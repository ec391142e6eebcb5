//! The reference's expectations for the `BWT` trait (rust-msbwt src/rle_bwt.rs:603-710,
//! src/msbwt_core.rs:110-122, src/lib.rs:25-28), asked of `GpuRleBWT`, plus GPU-vs-CPU agreement on
//! batches.  Needs an MI355X and MSBWT_HIP_LIB_DIR; `cargo test` from this directory.
use msbwt2::bwt_converter::convert_to_vec;
use msbwt2::bwt_util::naive_bwt;
use msbwt2::msbwt_core::{BWTRange, BWT};
use msbwt2::rle_bwt::RleBWT;
use msbwt2::string_util::convert_stoi;
use msbwt2_hip::GpuRleBWT;

fn both(strings: &[&str]) -> (RleBWT, GpuRleBWT) {
    let rle = convert_to_vec(naive_bwt(strings).as_bytes());
    let (mut cpu, mut gpu) = (RleBWT::new(), GpuRleBWT::new());
    cpu.load_vector(rle.clone());
    gpu.load_vector(rle);
    (cpu, gpu)
}

#[test]
fn totals_and_literal_counts() {
    let strings = ["CCGTACGTA", "GGTACAGTA", "ACGACGACG"];
    let (cpu, gpu) = both(&strings);
    assert_eq!(gpu.get_total_size(), cpu.get_total_size());
    for sym in 0..6u8 {
        assert_eq!(gpu.get_symbol_count(sym), cpu.get_symbol_count(sym));
        assert_eq!(gpu.count_kmer(&[sym]), cpu.get_symbol_count(sym));
    }
    for s in strings.iter() {
        assert_eq!(gpu.count_kmer(&convert_stoi(s)), 1);
    }
    assert_eq!(gpu.count_kmer(&convert_stoi("ACG")), 4);
    assert_eq!(gpu.count_kmer(&convert_stoi("CC")), 1);
    assert_eq!(gpu.count_kmer(&convert_stoi("TAC")), 2);
    assert_eq!(gpu.count_kmer(&[]), gpu.get_total_size());
}

#[test]
fn constrain_range_agrees_everywhere() {
    let (cpu, gpu) = both(&["CCGT", "N", "ACG"]);
    let total = cpu.get_total_size();
    for sym in 0..6u8 {
        for l in 0..=total {
            for h in l..=total {
                let r = BWTRange { l, h };
                let (a, b) = unsafe { (cpu.constrain_range(sym, &r), gpu.constrain_range(sym, &r)) };
                assert_eq!((a.l, a.h), (b.l, b.h));
            }
        }
    }
}

#[test]
fn batches_equal_single_queries() {
    let (cpu, gpu) = both(&["ACGTTGCAACGT", "TTGACCA", "GATTACA", "ACGT"]);
    let k = 3;
    let mut kmers = Vec::new();
    for a in 1..6u8 { for b in 0..6u8 { for c in 1..6u8 { kmers.extend_from_slice(&[a, b, c]); } } }
    let got = gpu.count_kmers(&kmers, k);
    for (i, q) in kmers.chunks(k).enumerate() {
        assert_eq!(got[i], cpu.count_kmer(q));
    }
}

#[test]
fn two_string_fixture() {
    // test_data/two_string.npy of rust-msbwt: count("ACGT") == 1
    let path = std::env::var("MSBWT_TWO_STRING_NPY").unwrap_or_else(|_| "test_data/two_string.npy".to_string());
    let mut gpu = GpuRleBWT::new();
    gpu.load_numpy_file(&path).unwrap();
    assert_eq!(gpu.count_kmer(&convert_stoi("ACGT")), 1);
    assert_eq!(gpu.count_kmer(&convert_stoi("TGCA")), 1);
    assert!(gpu.load_numpy_file("/nonexistent/file.npy").is_err());
}

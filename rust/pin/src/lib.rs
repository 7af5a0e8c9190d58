//! Nothing to build: this package exists for `cargo test` (tests/golden.rs -> ../../tests/golden.rs).

// Small functions in the Rust subset tools/rust_air_eval.py interprets.  tests/test_rust_interp.py runs them and compares with the values
// the Rust language defines for them (worked out by hand, noted beside each function).  They pin the interpreter's handling of the
// constructs the reference's code relies on: by-value scalars behind `&mut`, `*x = array`, element references from `iter_mut` / `&mut xs`,
// closures that assign captured variables, `chunks_mut`, `?`, Option from `bool::then`, operator traits, integer widths.

pub fn bump(counter: &mut usize) {
    *counter += 1;
}

// -> 3
pub fn mut_scalar_through_calls() -> usize {
    let mut c = 0;
    bump(&mut c);
    bump(&mut c);
    bump(&mut c);
    c
}

fn replace(state: &mut [u64; 4]) {
    *state = [state[3], state[2], state[1], state[0]];
}

// -> [4, 3, 2, 1]
pub fn deref_assign_array() -> [u64; 4] {
    let mut s = [1, 2, 3, 4];
    replace(&mut s);
    s
}

// -> [11, 22, 33]
pub fn iter_mut_zip() -> Vec<u64> {
    let mut acc = vec![1, 2, 3];
    let add = vec![10, 20, 30];
    for (&a, x) in add.iter().zip(&mut acc) {
        *x += a;
    }
    acc
}

// -> [1, 20, 30, 4]
pub fn iter_mut_skip_take() -> Vec<u64> {
    let mut v = vec![1, 2, 3, 4];
    for e in v.iter_mut().skip(1).take(2) {
        *e = *e * 10;
    }
    v
}

// -> [0, 0, 1, 3, 6]   (the closure keeps its running sum between calls)
pub fn closure_assigns_captured() -> Vec<usize> {
    let mut running = 0;
    (0..5)
        .map(|i| {
            let before = running;
            running += i;
            before
        })
        .collect()
}

// -> [1, 1, 1, 1, 12, 12, 12, 12]
pub fn chunks_mut_for_each() -> Vec<usize> {
    let mut v = vec![0; 8];
    v.chunks_mut(4).enumerate().for_each(|(i, chunk)| {
        for x in chunk.iter_mut() {
            *x = 1 + 11 * i;
        }
    });
    v
}

// -> 7   (assignment to a variable of the enclosing block from a nested else-branch)
pub fn nested_assignment(n: usize) -> usize {
    let mut result = 0;
    if n > 100 {
        result = 1;
    } else {
        if n > 10 {
            result = 2;
        } else {
            result = 7;
        }
    }
    result
}

fn may_fail(x: usize) -> Result<usize> {
    ensure!(x < 3, "too big");
    Ok(x + 1)
}

fn try_twice(x: usize) -> Result<usize> {
    let a = may_fail(x)?;
    let b = may_fail(a)?;
    Ok(b)
}

// try_twice(0) -> Ok(2); try_twice(2) -> Err (the second call fails and `?` returns it)
pub fn question_mark(x: usize) -> Result<usize> {
    try_twice(x)
}

// uses_option(true) -> 3 (the vector's length), uses_option(false) -> 0
pub fn uses_option(flag: bool) -> usize {
    let v = flag.then(|| vec![5, 6, 7]);
    v.as_ref().map(|xs| xs.len()).unwrap_or(0)
}

fn bytes_of(x: u32) -> Vec<u8> {
    x.to_le_bytes().to_vec()
}

// -> [0x78, 0x56, 0x34, 0x12]   (the width comes from the parameter's type, as in serialization.rs write_u32)
pub fn le_bytes() -> Vec<u8> {
    bytes_of(0x12345678)
}

// -> 0xEF  (truncating cast), 40 (leading zeros of a u64 holding 2^23)
pub fn casts() -> (u8, u32) {
    let a = 0xBEEFu64 as u8;
    let b = (1u64 << 23).leading_zeros();
    (a, b)
}

// -> 5   (x_index >>= 1 three times from 40)
pub fn shift_assign() -> usize {
    let mut x = 40;
    for _ in 0..3 {
        x >>= 1;
    }
    x
}

pub struct Sponge {
    state: [u64; 3],
}

impl Sponge {
    fn snapshot(&mut self) -> [u64; 3] {
        self.state
    }

    fn absorb(&mut self, x: u64) {
        self.state[0] = x;
    }
}

// -> ([1, 2, 3], [9, 2, 3])   (an array returned from a borrowed struct is a copy: the later write does not show in it)
pub fn array_leaves_by_copy() -> ([u64; 3], [u64; 3]) {
    let mut s = Sponge { state: [1, 2, 3] };
    let before = s.snapshot();
    s.absorb(9);
    (before, s.snapshot())
}

// -> 3   (loop with break through return; `loop` + early return)
pub fn loop_until() -> usize {
    let mut out = Vec::new();
    let mut i = 0;
    loop {
        out.push(i);
        if out.len() == 3 {
            return out.len();
        }
        i += 1;
    }
}

// -> [0, 4, 2, 6, 1, 5, 3, 7]   (swap-based bit reversal of 0..8, as cfft's permute)
pub fn swaps() -> Vec<usize> {
    let mut v: Vec<usize> = (0..8).collect();
    let n = v.len();
    for i in 0..n {
        let j = i.reverse_bits() >> (usize::BITS - 3);
        if j > i {
            v.swap(i, j);
        }
    }
    v
}

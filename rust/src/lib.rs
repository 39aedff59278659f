//! `vfind` Python module over the MI355X engine's C ABI (include/vfind_hip.h).
//! Same `find_variants` signature and defaults as the reference module; the per-read loop, the
//! parasail aligner and the HashMap reduce are gone: they run as HIP kernels behind `vf_*`.
use polars::prelude::*;
use polars_arrow::ffi;
use pyo3::exceptions::{PyMemoryError, PyOSError, PyRuntimeError, PyValueError};
use pyo3::prelude::*;
use pyo3_polars::PyDataFrame;
use std::ffi::{c_int, CStr, CString};
use std::fs::File;

mod vf_abi;
use vf_abi::*;

/// VF_EINVAL -> ValueError (same text as the reference's preflight), VF_EIO -> OSError, VF_ENOMEM -> MemoryError,
/// else RuntimeError (the codes of include/vfind_hip.h; tests/test_abi.py keeps these arms and vfind_amd/_lib.py::check in step)
fn check(rc: c_int) -> PyResult<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(vf_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        1 => PyValueError::new_err(msg),
        2 => PyOSError::new_err(msg),
        6 => PyMemoryError::new_err(msg),
        _ => PyRuntimeError::new_err(msg),
    })
}

struct Ctx(*mut VfCtx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { vf_ctx_destroy(self.0) }
    }
}

#[pyfunction]
#[pyo3(signature = (
    fq_path,
    adapters,
    match_score=3,
    mismatch_score=-2,
    gap_open_penalty=5,
    gap_extend_penalty=2,
    accept_prefix_alignment=0.75,
    accept_suffix_alignment=0.75,
    n_threads=3,
    queue_len=2,
    skip_translation=false,
    show_progress=true,
))]
#[allow(clippy::too_many_arguments)]
pub fn find_variants(
    py: Python<'_>,
    fq_path: String,
    adapters: (String, String),
    match_score: i32,
    mismatch_score: i32,
    gap_open_penalty: i32,
    gap_extend_penalty: i32,
    accept_prefix_alignment: f64,
    accept_suffix_alignment: f64,
    n_threads: u32,
    queue_len: usize,
    skip_translation: bool,
    show_progress: bool,
) -> PyResult<PyDataFrame> {
    let _ = show_progress; // the reference's spinner is never ticked
    File::open(&fq_path)?; // same error precedence as the reference: the open comes first
    let params = VfParams {
        prefix: adapters.0.as_ptr(),
        prefix_len: adapters.0.len() as u32,
        suffix: adapters.1.as_ptr(),
        suffix_len: adapters.1.len() as u32,
        match_score,
        mismatch_score,
        gap_open_penalty,
        gap_extend_penalty,
        accept_prefix_alignment,
        accept_suffix_alignment,
        n_threads,
        queue_len: queue_len as u32,
        skip_translation: skip_translation as i32,
        device: 0,
        table_slots: 0,
        arena_bytes: 0,
        batch_reads: 0,
        reserved: 0,
    };
    let mut raw = std::ptr::null_mut();
    check(unsafe { vf_ctx_create(&params, &mut raw) })?;
    let ctx = Ctx(raw);
    let path = CString::new(fq_path).map_err(|e| PyValueError::new_err(e.to_string()))?;

    // decode + H2D + kernels run without the GIL (the reference holds it for the whole call)
    let rc = py.allow_threads(|| unsafe { vf_run_fastq(ctx.0, path.as_ptr()) });
    check(rc)?;
    let mut res = std::ptr::null_mut();
    check(unsafe { vf_finish(ctx.0, &mut res) })?;

    // {sequence: large_utf8, count: uint64} through the Arrow C Data Interface, zero copy
    let mut array = ffi::ArrowArray::empty();
    let mut schema = ffi::ArrowSchema::empty();
    let rc = unsafe { vf_result_export_arrow(res, &mut array, &mut schema) };
    unsafe { vf_result_free(res) };
    check(rc)?;
    let field = unsafe { ffi::import_field_from_c(&schema) }.map_err(|e| PyValueError::new_err(e.to_string()))?;
    let chunk = unsafe { ffi::import_array_from_c(array, field.dtype.clone()) }.map_err(|e| PyValueError::new_err(e.to_string()))?;
    let s = Series::from_arrow("frame".into(), chunk).map_err(|e| PyValueError::new_err(format!("Error creating DataFrame: {}", e)))?;
    let df = s
        .struct_()
        .map_err(|e| PyValueError::new_err(format!("Error creating DataFrame: {}", e)))?
        .clone()
        .unnest();
    Ok(PyDataFrame(df))
}

/// vFind Python module
#[pymodule]
fn vfind(m: &Bound<'_, PyModule>) -> PyResult<()> {
    m.add_function(wrap_pyfunction!(find_variants, m)?)?;
    Ok(())
}

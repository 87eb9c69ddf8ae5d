;;;; bindings.lisp -- CFFI declarations of include/mhx.h, one defcfun per entry point.
(in-package #:mcmc-fitting-amd)

(cffi:define-foreign-library libmhx
  (:unix (:or "libmhx.so" "./libmhx.so"))
  (t (:default "libmhx")))

(defun load-libmhx (&optional path)
  "Load libmhx.so (PATH overrides the search).  There is no pure-Lisp fallback: without the
library, or without a gfx950 device, walker-create signals MHX-ERROR."
  (if path
      (cffi:load-foreign-library path)
      (cffi:use-foreign-library libmhx)))

(define-condition mhx-error (error)
  ((code :initarg :code :reader mhx-error-code)
   (message :initarg :message :reader mhx-error-message))
  (:report (lambda (c s)
             (format s "libmhx error ~d: ~a" (mhx-error-code c) (mhx-error-message c)))))

;;; status codes / enums of mhx.h
(defconstant +mhx-ok+ 0)
(defconstant +lik-normal+ 0)
(defconstant +lik-normal-cutoff+ 1)
(defconstant +lik-poisson+ 2)
(defconstant +lik-expr+ 3)
(defconstant +chain-running+ 0)
(defconstant +chain-done+ 1)
(defconstant +chain-fp-trap+ 2)
(defconstant +chain-stopped+ 3)

(cffi:defcstruct mhx-config
  (n-chains :int64)
  (n-params :int32)
  (n-functions :int32)
  (device :int32)
  (adapt-mode :int32)
  (seed :uint64)
  (chain-offset :int64)
  (history-capacity :int32)
  (poisson-logfact-double :int32))

(cffi:defcstruct mhx-run-opts
  (n :int64)
  (temperature :double)
  (auto-mode :int32)
  (max-walker-length :int64)
  (l-matrix :pointer)
  (l-matrix-per-chain :int32))

(cffi:defcfun ("mhx_version" %mhx-version) :int)
(cffi:defcfun ("mhx_build_id" %mhx-build-id) :string)
(cffi:defcfun ("mhx_last_error" %mhx-last-error) :string)
(cffi:defcfun ("mhx_device_count" %mhx-device-count) :int (count :pointer))
(cffi:defcfun ("mhx_create" %mhx-create) :int (cfg :pointer) (out :pointer))
(cffi:defcfun ("mhx_destroy" %mhx-destroy) :void (e :pointer))
(cffi:defcfun ("mhx_set_function" %mhx-set-function) :int
  (e :pointer) (k :int) (model-id :int) (shape :pointer) (n-shape :int)
  (param-index :pointer) (n-index :int))
(cffi:defcfun ("mhx_set_dataset" %mhx-set-dataset) :int
  (e :pointer) (k :int) (x :pointer) (y :pointer) (sigma :pointer) (n :size) (likelihood :int))
(cffi:defcfun ("mhx_set_dataset_cols" %mhx-set-dataset-cols) :int
  (e :pointer) (k :int) (xcols :pointer) (n-cols :int) (y :pointer) (sigma :pointer) (n :size)
  (likelihood :int))
(cffi:defcfun ("mhx_set_bounds" %mhx-set-bounds) :int
  (e :pointer) (k :int) (idx :pointer) (lo :pointer) (hi :pointer) (n :int))
(cffi:defcfun ("mhx_set_function_expr" %mhx-set-function-expr) :int
  (e :pointer) (k :int) (expr :string) (param-names :pointer) (param-index :pointer) (n-index :int))
(cffi:defcfun ("mhx_set_expr_recognition" %mhx-set-expr-recognition) :int (e :pointer) (on :int))
(cffi:defcfun ("mhx_expr_classify" %mhx-expr-classify) :int
  (expr :string) (param-names :pointer) (n-names :int) (model :pointer) (shape :pointer)
  (order :pointer) (n-order :pointer))
(cffi:defcfun ("mhx_set_prior_expr" %mhx-set-prior-expr) :int
  (e :pointer) (k :int) (expr :string) (names :pointer) (index :pointer) (n :int))
(cffi:defcfun ("mhx_set_likelihood_expr" %mhx-set-likelihood-expr) :int
  (e :pointer) (k :int) (expr :string))
(cffi:defcfun ("mhx_walker_modify" %mhx-walker-modify) :int (e :pointer) (action :int) (n :int64))
(cffi:defcfun ("mhx_set_history" %mhx-set-history) :int
  (e :pointer) (chain :int64) (prob :pointer) (theta :pointer) (n :int))
(cffi:defcfun ("mhx_get_pooled" %mhx-get-pooled) :int
  (e :pointer) (stats :pointer) (l-pool :pointer) (valid :pointer) (refreshes :pointer))
(cffi:defcfun ("mhx_init_chains" %mhx-init-chains) :int
  (e :pointer) (theta0 :pointer) (broadcast :int))
(cffi:defcfun ("mhx_logpost" %mhx-logpost) :int
  (e :pointer) (theta :pointer) (n :size) (out :pointer) (parts :pointer))
(cffi:defcfun ("mhx_step_injected" %mhx-step-injected) :int
  (e :pointer) (l :pointer) (per-chain-l :int) (z :pointer) (u :pointer) (temp :pointer)
  (accepted-out :pointer))
(cffi:defcfun ("mhx_run_opts_default" %mhx-run-opts-default) :void (o :pointer))
(cffi:defcfun ("mhx_adaptive_begin" %mhx-adaptive-begin) :int (e :pointer) (o :pointer))
(cffi:defcfun ("mhx_adaptive_advance" %mhx-adaptive-advance) :int
  (e :pointer) (max-iters :int64) (n-running :pointer))
(cffi:defcfun ("mhx_adaptive_steps_full" %mhx-adaptive-steps-full) :int (e :pointer) (o :pointer))
(cffi:defcfun ("mhx_adaptive_steps" %mhx-adaptive-steps) :int (e :pointer) (n :int64))
(cffi:defcfun ("mhx_many_steps" %mhx-many-steps) :int
  (e :pointer) (n :int64) (l :pointer) (per-chain-l :int))
(cffi:defcfun ("mhx_request_stop" %mhx-request-stop) :int (e :pointer))
(cffi:defcfun ("mhx_set_allreduce" %mhx-set-allreduce) :int
  (e :pointer) (fn :pointer) (ctx :pointer) (wants-device-buffer :int))
(cffi:defcfun ("mhx_get_state" %mhx-get-state) :int
  (e :pointer) (theta :pointer) (logpost :pointer) (best-theta :pointer) (best-logpost :pointer)
  (length :pointer) (age :pointer))
(cffi:defcfun ("mhx_get_chain_status" %mhx-get-chain-status) :int
  (e :pointer) (status :pointer) (loop-index :pointer))
(cffi:defcfun ("mhx_get_lmatrix" %mhx-get-lmatrix) :int (e :pointer) (l :pointer))
(cffi:defcfun ("mhx_get_temperature" %mhx-get-temperature) :int (e :pointer) (temp :pointer))
(cffi:defcfun ("mhx_get_acceptance" %mhx-get-acceptance) :int
  (e :pointer) (take :int) (out :pointer))
(cffi:defcfun ("mhx_get_trace" %mhx-get-trace) :int
  (e :pointer) (chain :int64) (take :int) (prob :pointer) (theta :pointer) (n-out :pointer))
(cffi:defcfun ("mhx_get_proposal_factor" %mhx-get-proposal-factor) :int
  (e :pointer) (chain :int64) (take :int) (l-out :pointer) (status :pointer) (n-forward :pointer))
(cffi:defcfun ("mhx_get_counters" %mhx-get-counters) :int
  (e :pointer) (chain-steps :pointer) (kernel-launches :pointer))
(cffi:defcfun ("mhx_kernel_name" %mhx-kernel-name) :string (e :pointer))
(cffi:defcfun ("mhx_kernel_timing" %mhx-kernel-timing) :int
  (e :pointer) (reset :int) (avg-ms :pointer) (launches :pointer) (total-ms :pointer))

(cffi:defcfun ("mhx_take_step" %mhx-take-step) :int
  (e :pointer) (l :pointer) (per-chain-l :int) (temperature :double))
(cffi:defcfun ("mhx_get_chain" %mhx-get-chain) :int
  (e :pointer) (chain :int64) (theta :pointer) (logpost :pointer) (best-theta :pointer)
  (best-logpost :pointer) (length :pointer) (age :pointer))
;;; native RCCL for one-process-per-GPU hosts (a Lisp image per GPU, e.g. under MPI)
(cffi:defcfun ("mhx_comm_get_unique_id" %mhx-comm-get-unique-id) :int (id :pointer))
(cffi:defcfun ("mhx_comm_init_rank" %mhx-comm-init-rank) :int
  (e :pointer) (id :pointer) (rank :int) (n-ranks :int))
;;; one Lisp image, several GPUs
(cffi:defcfun ("mhx_group_partition" %mhx-group-partition) :int
  (n-chains :int64) (n-parts :int) (part :int) (first :pointer) (count :pointer))
(cffi:defcfun ("mhx_group_create" %mhx-group-create) :int
  (cfg :pointer) (devices :pointer) (n-devices :int) (out :pointer))
(cffi:defcfun ("mhx_group_destroy" %mhx-group-destroy) :void (g :pointer))
(cffi:defcfun ("mhx_group_size" %mhx-group-size) :int (g :pointer))
(cffi:defcfun ("mhx_group_engine" %mhx-group-engine) :pointer (g :pointer) (i :int))
(cffi:defcfun ("mhx_group_chain_range" %mhx-group-chain-range) :int
  (g :pointer) (i :int) (first :pointer) (count :pointer))
(cffi:defcfun ("mhx_group_set_function" %mhx-group-set-function) :int
  (g :pointer) (k :int) (model-id :int) (shape :pointer) (n-shape :int)
  (param-index :pointer) (n-index :int))
(cffi:defcfun ("mhx_group_set_dataset" %mhx-group-set-dataset) :int
  (g :pointer) (k :int) (x :pointer) (y :pointer) (sigma :pointer) (n :size) (likelihood :int))
(cffi:defcfun ("mhx_group_set_dataset_cols" %mhx-group-set-dataset-cols) :int
  (g :pointer) (k :int) (xcols :pointer) (n-cols :int) (y :pointer) (sigma :pointer) (n :size)
  (likelihood :int))
(cffi:defcfun ("mhx_group_set_bounds" %mhx-group-set-bounds) :int
  (g :pointer) (k :int) (idx :pointer) (lo :pointer) (hi :pointer) (n :int))
(cffi:defcfun ("mhx_group_set_function_expr" %mhx-group-set-function-expr) :int
  (g :pointer) (k :int) (expr :string) (param-names :pointer) (param-index :pointer) (n-index :int))
(cffi:defcfun ("mhx_group_set_expr_recognition" %mhx-group-set-expr-recognition) :int
  (g :pointer) (on :int))
(cffi:defcfun ("mhx_group_set_prior_expr" %mhx-group-set-prior-expr) :int
  (g :pointer) (k :int) (expr :string) (names :pointer) (index :pointer) (n :int))
(cffi:defcfun ("mhx_group_set_likelihood_expr" %mhx-group-set-likelihood-expr) :int
  (g :pointer) (k :int) (expr :string))
(cffi:defcfun ("mhx_group_init_chains" %mhx-group-init-chains) :int
  (g :pointer) (theta0 :pointer) (broadcast :int))
(cffi:defcfun ("mhx_group_adaptive_begin" %mhx-group-adaptive-begin) :int (g :pointer) (o :pointer))
(cffi:defcfun ("mhx_group_adaptive_advance" %mhx-group-adaptive-advance) :int
  (g :pointer) (max-iters :int64) (n-running :pointer))
(cffi:defcfun ("mhx_group_adaptive_steps_full" %mhx-group-adaptive-steps-full) :int
  (g :pointer) (o :pointer))
(cffi:defcfun ("mhx_group_request_stop" %mhx-group-request-stop) :int (g :pointer))
(cffi:defcfun ("mhx_group_get_state" %mhx-group-get-state) :int
  (g :pointer) (theta :pointer) (logpost :pointer) (best-theta :pointer) (best-logpost :pointer)
  (length :pointer) (age :pointer))
(cffi:defcfun ("mhx_group_get_counters" %mhx-group-get-counters) :int
  (g :pointer) (chain-steps :pointer) (kernel-launches :pointer))

(defmacro with-c-call (&body body)
  "HIP/RCCL runtime code may raise inexact/invalid flags that SBCL turns into conditions;
mask the traps around every foreign call (SURVEY 8b)."
  #+sbcl `(sb-int:with-float-traps-masked (:invalid :divide-by-zero :overflow :inexact)
            ,@body)
  #-sbcl `(progn ,@body))

(defun check (rc)
  (unless (= rc +mhx-ok+)
    (error 'mhx-error :code rc :message (%mhx-last-error)))
  rc)

(defun fill-doubles (ptr seq)
  (let ((i 0))
    (map nil (lambda (v)
               (setf (cffi:mem-aref ptr :double i) (coerce v 'double-float))
               (incf i))
         seq)
    ptr))

(defun fill-int32s (ptr seq)
  (let ((i 0))
    (map nil (lambda (v) (setf (cffi:mem-aref ptr :int32 i) v) (incf i)) seq)
    ptr))

(defmacro with-c-strings ((var strings) &body body)
  "VAR: a foreign array of char* holding copies of STRINGS for the extent of BODY"
  (let ((n (gensym "N")) (ptrs (gensym "PTRS")) (i (gensym "I")))
    `(let* ((,n (length ,strings))
            (,ptrs (mapcar #'cffi:foreign-string-alloc ,strings)))
       (unwind-protect
            (cffi:with-foreign-object (,var :pointer (max 1 ,n))
              (loop for p in ,ptrs for ,i from 0
                    do (setf (cffi:mem-aref ,var :pointer ,i) p))
              ,@body)
         (mapc #'cffi:foreign-string-free ,ptrs)))))

(defun read-doubles (ptr n)
  (let ((out (make-array n :element-type 'double-float)))
    (dotimes (i n out)
      (setf (aref out i) (cffi:mem-aref ptr :double i)))))

;;;; walker.lisp -- walker-create / walker-adaptive-steps / walker-get over libmhx.
;;; Mirrors the reference's surface for the walker-adaptive-steps path; every number is
;;; computed below the C ABI.  Citations M: = mcmc-fitting.lisp of the reference.
(in-package #:mcmc-fitting-amd)

(defvar mfit-walker-estop nil
  "Kept for source compatibility (M:860-861).  Setting it from another thread has no effect
on a running kernel; call (request-stop walker) instead, which raises the device flag the
kernel polls at the top of every loop iteration.")

(defstruct walker-step
  (prob most-negative-double-float :type float)
  (params nil :type list))

(defstruct walker
  (engine (cffi:null-pointer))
  (function nil :type list)
  (param-keys nil :type list)
  (param-style :multiple-kwargs :type symbol)
  (data nil :type list)
  (data-error nil :type list)
  (log-liklihood nil :type list)
  (log-prior nil :type list)
  (n-chains 1 :type integer)
  (n-params 0 :type integer))

(defun force-list (item) (if (consp item) item (list item))) ; M:755-759

(defun get-depth (tree)                 ; M:761-772
  (cond ((null tree) nil)
        ((numberp tree) 0)
        ((arrayp tree) (array-rank tree))
        (t (+ 1 (get-depth (elt tree 0))))))

(defun clean-data (data number-of-functions) ; M:807-825
  (cond ((= (get-depth data) 1)
         (error "clean-data: data is of insufficient depth or improperly structured."))
        ((= (get-depth data) 2) (clean-data (list data) number-of-functions))
        ((= (length data) number-of-functions)
         (mapcar (lambda (ds) (mapcar (lambda (col) (coerce col 'list)) ds)) data))
        (t (error "clean-data: insufficient number of datasets, ~a, for the given number of functions, ~a."
                  (length data) number-of-functions))))

(defun clean-data-error (stddev ys)
  "M:774-805 for the layouts the path uses: a number broadcasts; a structure equal to the y
structure is taken as is; anything else broadcasts its first element."
  (labels ((first-element (tree)
             (cond ((null tree) nil)
                   ((numberp tree) tree)
                   (t (first-element (elt tree 0))))))
    (cond ((numberp stddev)
           (mapcar (lambda (y) (make-list (length y) :initial-element stddev)) ys))
          ((and (= (length stddev) (length ys))
                (every (lambda (s y) (and (not (numberp s)) (= (length s) (length y))))
                       stddev ys))
           (mapcar (lambda (s) (coerce s 'list)) stddev))
          ((and (= (length ys) 1) (every #'numberp stddev)
                (= (length stddev) (length (first ys))))
           (list (coerce stddev 'list)))
          (t (let ((v (first-element stddev)))
               (mapcar (lambda (y) (make-list (length y) :initial-element v)) ys))))))

(defun plist-keys (plist)               ; M:190-193, first occurrence wins
  (let ((keys nil))
    (loop for (k nil) on plist by #'cddr
          do (unless (member k keys) (push k keys)))
    (nreverse keys)))

(defun engine-of (walker)
  (let ((e (walker-engine walker)))
    (when (cffi:null-pointer-p e) (error "walker has been destroyed"))
    e))

(defun signal-if-trapped (walker)
  "A frozen chain is where the reference would have signalled an unhandled float trap."
  (let* ((e (engine-of walker)) (n (walker-n-chains walker)))
    (cffi:with-foreign-object (st :int32 n)
      (with-c-call (check (%mhx-get-chain-status e st (cffi:null-pointer))))
      (dotimes (c n)
        (when (= (cffi:mem-aref st :int32 c) +chain-fp-trap+)
          (error 'floating-point-invalid-operation
                 :operation 'walker-take-step :operands (list :chain c)))))))

(defun walker-create (&key function data params data-error log-liklihood log-prior param-bounds
                        (n-chains 1) (device 0) (seed 0) (chain-offset 0) (history-capacity 0))
  "(walker-create &key function data params data-error log-liklihood log-prior param-bounds)
M:1132-1163.  :function takes model designators (models.lisp); everything else as the
reference, single items or lists with one item per function.  N-CHAINS > 1 makes a walker
set that steps as one batch."
  (declare (ignorable param-bounds))
  (let* ((function (force-list function))
         (k-fns (length function))
         (data (clean-data data k-fns))
         (ys (mapcar #'second data))
         (data-error (clean-data-error (if data-error data-error 1) ys))
         (keys (plist-keys params))
         (d (length keys))
         (values (mapcar (lambda (k) (coerce (getf params k) 'double-float)) keys))
         (liks (if (consp log-liklihood) log-liklihood
                   (make-list k-fns :initial-element log-liklihood)))
         (pris (if (consp log-prior) log-prior (make-list k-fns :initial-element log-prior)))
         (engine (cffi:null-pointer)))
    (unless (every #'model-p function)
      (error 'mhx-error :code -5 :message ":function must be a model designator (see models.lisp)"))
    (cffi:with-foreign-objects ((cfg '(:struct mhx-config)) (out :pointer))
      (dotimes (i (cffi:foreign-type-size '(:struct mhx-config)))
        (setf (cffi:mem-aref cfg :uint8 i) 0))
      (setf (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-chains) n-chains
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-params) d
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-functions) k-fns
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'device) device
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'seed) seed
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'chain-offset) chain-offset
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'history-capacity) history-capacity)
      (with-c-call (check (%mhx-create cfg out)))
      (setf engine (cffi:mem-ref out :pointer)))
    (let ((ok nil))
     (unwind-protect
      (progn
      (loop for fn in function
            for k from 0
            for ds in data
            for sg in data-error
            for lik in liks
            for pri in pris
            do (let* ((idx (mapcar (lambda (key)
                                     (or (position key keys)
                                         (error "function ~d reads key ~s that :params does not supply"
                                                k key)))
                                   (model-keys fn)))
                      (shape (model-shape fn))
                      (n (length (first ds))))
                 (cffi:with-foreign-objects ((cidx :int32 (max 1 (length idx)))
                                             (cshape :int32 (max 1 (length shape))))
                   (fill-int32s cidx idx)
                   (fill-int32s cshape shape)
                   (if (model-expr fn)
                       ;; an arbitrary closure body: compiled for gfx950 at init (expr.lisp)
                       (with-c-strings (cnames (mapcar #'mangle-symbol (model-keys fn)))
                         (with-c-call
                           (check (%mhx-set-function-expr engine k (model-expr fn) cnames cidx
                                                          (length idx)))))
                       (with-c-call
                         (check (%mhx-set-function engine k (model-id fn) cshape (length shape)
                                                   cidx (length idx))))))
                 (cffi:with-foreign-objects ((cx :double (max 1 n)) (cy :double (max 1 n))
                                             (cs :double (max 1 n)))
                   (fill-doubles cx (first ds))
                   (fill-doubles cy (second ds))
                   (fill-doubles cs sg)
                   (with-c-call
                     (check (%mhx-set-dataset engine k cx cy cs n (likelihood-id lik)))))
                 (when (likelihood-spec-p lik)
                   (with-c-call
                     (check (%mhx-set-likelihood-expr engine k (likelihood-spec-expr lik)))))
                 (let ((bounds (cond ((null pri) nil)
                                     ((eq pri 'log-prior-flat) nil)
                                     ((eq pri #'log-prior-flat) nil)
                                     ((prior-bounds-spec-p pri) (prior-bounds-spec-bounds pri))
                                     (t (error 'mhx-error :code -5
                                                          :message ":log-prior must be log-prior-flat or (prior-bounds ...)")))))
                   (cffi:with-foreign-objects ((bi :int32 (max 1 (length bounds)))
                                               (lo :double (max 1 (length bounds)))
                                               (hi :double (max 1 (length bounds))))
                     (fill-int32s bi (mapcar (lambda (b) (or (position (first b) keys) -1)) bounds))
                     (fill-doubles lo (mapcar #'second bounds))
                     (fill-doubles hi (mapcar #'third bounds))
                     (with-c-call
                       (check (%mhx-set-bounds engine k bi lo hi (length bounds)))))
                   (when (and (prior-bounds-spec-p pri) (prior-bounds-spec-body-expr pri))
                     (cffi:with-foreign-object (gi :int32 (max 1 d))
                       (fill-int32s gi (loop for i below d collect i))
                       (with-c-strings (cnames (mapcar #'mangle-symbol keys))
                         (with-c-call
                           (check (%mhx-set-prior-expr engine k (prior-bounds-spec-body-expr pri)
                                                       cnames gi d)))))))))
      (cffi:with-foreign-object (th :double d)
        (fill-doubles th values)
        (with-c-call (check (%mhx-init-chains engine th 1))))
      (setf ok t))
      (unless ok (with-c-call (%mhx-destroy engine)))))
    (let ((w (make-walker :engine engine :function function :param-keys keys
                          :data data :data-error data-error
                          :log-liklihood liks :log-prior pris
                          :n-chains n-chains :n-params d)))
      (signal-if-trapped w)
      w)))

(defun walker-destroy (walker)
  (unless (cffi:null-pointer-p (walker-engine walker))
    (with-c-call (%mhx-destroy (walker-engine walker)))
    (setf (walker-engine walker) (cffi:null-pointer)))
  nil)

(defun request-stop (walker)
  "What (setf mfit-walker-estop t) does in the reference (M:860-861, polled at M:904)."
  (with-c-call (check (%mhx-request-stop (engine-of walker)))))

;;; ------------------------------------------------------------------ state read-back
(defun %state (walker chain)
  "values: theta prob best-theta best-prob length age  (chain CHAIN)"
  (let* ((e (engine-of walker)) (n (walker-n-chains walker)) (d (walker-n-params walker)))
    (cffi:with-foreign-objects ((th :double (* n d)) (lp :double n) (bt :double (* n d))
                                (bl :double n) (ln :int64 n) (ag :int64 n))
      (with-c-call (check (%mhx-get-state e th lp bt bl ln ag)))
      (values (read-doubles (cffi:inc-pointer th (* 8 chain d)) d)
              (cffi:mem-aref lp :double chain)
              (read-doubles (cffi:inc-pointer bt (* 8 chain d)) d)
              (cffi:mem-aref bl :double chain)
              (cffi:mem-aref ln :int64 chain)
              (cffi:mem-aref ag :int64 chain)))))

(defun %plist (walker vec)
  (loop for k in (walker-param-keys walker)
        for i from 0
        append (list k (aref vec i))))

(defun walker-last-step (walker &optional (chain 0))
  (multiple-value-bind (th lp) (%state walker chain)
    (make-walker-step :prob lp :params (%plist walker th))))

(defun walker-most-likely-step (walker &optional (chain 0))
  (multiple-value-bind (th lp bt bl) (%state walker chain)
    (declare (ignore th lp))
    (make-walker-step :prob bl :params (%plist walker bt))))

(defun walker-length (walker &optional (chain 0))
  (nth-value 4 (%state walker chain)))

(defun walker-age (walker &optional (chain 0))
  (nth-value 5 (%state walker chain)))

(defun walker-chain-status (walker)
  (let* ((e (engine-of walker)) (n (walker-n-chains walker)))
    (cffi:with-foreign-object (st :int32 n)
      (with-c-call (check (%mhx-get-chain-status e st (cffi:null-pointer))))
      (loop for c below n collect (cffi:mem-aref st :int32 c)))))

(defun walker-kernel-name (walker)
  "Which kernels serve this walker's problem, e.g. \"w8/lorder_normal\" or
\"w8/rtc[expr:normal] split x24\" (mhx_kernel_name)."
  (with-c-call (%mhx-kernel-name (engine-of walker))))

(defun %trace (walker chain take)
  "newest-first list of walker-steps, (walker-get :get :steps :take take)"
  (let* ((e (engine-of walker)) (d (walker-n-params walker))
         (take (max 1 take)))
    (cffi:with-foreign-objects ((pr :double take) (th :double (* take d)) (n-out :int))
      (with-c-call (check (%mhx-get-trace e chain take pr th n-out)))
      (loop for s below (cffi:mem-ref n-out :int)
            collect (make-walker-step
                     :prob (cffi:mem-aref pr :double s)
                     :params (%plist walker (read-doubles (cffi:inc-pointer th (* 8 s d)) d)))))))

(defun walker-walk (walker &optional (chain 0))
  (%trace walker chain (walker-length walker chain)))

(defun walker-modify (walker &key modify burn-number keep-number &allow-other-keys)
  "M:547-580.  :add-step happens on the device inside walker-take-step; :add-walks is unused
by the reference itself (M:556 discards the nconc)."
  (let ((e (engine-of walker)))
    (ecase modify
      (:burn-walks (with-c-call (check (%mhx-walker-modify e 0 burn-number))))
      (:keep-walks (with-c-call (check (%mhx-walker-modify e 1 keep-number))))
      (:reset (with-c-call (check (%mhx-walker-modify e 2 0))) walker)
      (:reset-to-most-likely (with-c-call (check (%mhx-walker-modify e 3 0))) walker)
      (:delete (walker-destroy walker)))))

;;; ------------------------------------------------------------------ stepping
(defun walker-adaptive-steps-full (walker &key (n 100000) (temperature 1d3)
                                            (auto :prob-settle)
                                            (sampling-optimization :covariance)
                                            max-walker-length l-matrix)
  "M:862.  :auto :slope-settle and :sampling-optimization :best-value are outside the
accelerated path and signal MHX-ERROR."
  (unless (eq sampling-optimization :covariance)
    (error 'mhx-error :code -5 :message ":best-value is outside the accelerated path"))
  (when (eq auto :slope-settle)
    (error 'mhx-error :code -5 :message ":slope-settle is outside the accelerated path"))
  (setf mfit-walker-estop nil)          ; M:865
  (let* ((e (engine-of walker)) (d (walker-n-params walker)))
    (cffi:with-foreign-objects ((o '(:struct mhx-run-opts)) (lm :double (* d d)))
      (%mhx-run-opts-default o)
      (setf (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'n) (floor n)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'temperature)
            (coerce temperature 'double-float)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'auto-mode) (if auto 1 0)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'max-walker-length)
            (if max-walker-length (floor max-walker-length) 0)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'l-matrix-per-chain) 0
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'l-matrix)
            (if l-matrix
                (progn (dotimes (i d)
                         (dotimes (j d)
                           (setf (cffi:mem-aref lm :double (+ (* i d) j))
                                 (coerce (aref l-matrix i j) 'double-float))))
                       lm)
                (cffi:null-pointer)))
      (with-c-call (check (%mhx-adaptive-steps-full e o)))))
  (signal-if-trapped walker)
  nil)

(defun walker-adaptive-steps (walker &optional (n 30000))
  "M:946-947"
  (walker-adaptive-steps-full walker :n n :temperature 10 :auto :prob-settle))

(defun mcmc-fit (&rest args &key function data params data-error log-liklihood log-prior
                              param-bounds &allow-other-keys)
  "M:1165-1175"
  (declare (ignore function data params data-error log-liklihood log-prior param-bounds))
  (let ((walker (apply #'walker-create args)))
    (walker-adaptive-steps walker)
    walker))

(defun %with-matrix (l-matrix d fn)
  (cffi:with-foreign-object (lm :double (* d d))
    (dotimes (i d)
      (dotimes (j d)
        (setf (cffi:mem-aref lm :double (+ (* i d) j)) (coerce (aref l-matrix i j) 'double-float))))
    (funcall fn lm)))

(defun walker-many-steps (walker n l-matrix)
  "M:849-853 with an explicit l-matrix (the nil default needs :median-params of the host)."
  (%with-matrix l-matrix (walker-n-params walker)
                (lambda (lm)
                  (with-c-call (check (%mhx-many-steps (engine-of walker) n lm 0)))))
  (signal-if-trapped walker)
  nil)

(defun walker-take-step (walker &key l-matrix (temperature 1) z u)
  "M:1072-1095 for a single-chain walker with the caller's randomness: Z = the d numbers
alexandria:gaussian-random would return (M:687), U = (random 1.0d0) (M:1092)."
  (let* ((e (engine-of walker)) (d (walker-n-params walker)))
    (unless (= (walker-n-chains walker) 1)
      (error "walker-take-step: single-chain walkers only; batches step through walker-adaptive-steps"))
    (%with-matrix
     l-matrix d
     (lambda (lm)
       (cffi:with-foreign-objects ((cz :double d) (cu :double) (ct :double) (acc :uint8))
         (fill-doubles cz z)
         (setf (cffi:mem-ref cu :double) (coerce u 'double-float)
               (cffi:mem-ref ct :double) (coerce temperature 'double-float))
         (with-c-call (check (%mhx-step-injected e lm 0 cz cu ct acc)))
         (signal-if-trapped walker)
         (= 1 (cffi:mem-ref acc :uint8)))))))

;;; ------------------------------------------------------------------ walker-get M:487-543
(defun median (sequence)                ; nth-percentile 50, M:1493-1517
  (let* ((copy (sort (copy-seq sequence) #'<))
         (n (* 50 (- (length copy) 1) 1/100)))
    (multiple-value-bind (pos rem) (floor n)
      (if (= rem 0)
          (elt copy pos)
          (/ (+ (elt copy pos) (elt copy (+ pos 1))) 2)))))

(defun walker-get (walker &key (get :steps) take param (chain 0))
  (let* ((e (engine-of walker))
         (len (walker-length walker chain))
         (take (if take (min len take) len))
         (d (walker-n-params walker)))
    (flet ((steps () (%trace walker chain take)))
      (case get
        (:steps (steps))
        (:log-liklihoods (mapcar #'walker-step-prob (steps)))
        (:params (mapcar #'walker-step-params (steps)))
        (:param (mapcar (lambda (s) (getf (walker-step-params s) param)) (steps)))
        (:unique-steps
         (mapcon (lambda (x)
                   (if (equal (walker-step-prob (car x))
                              (when (cadr x) (walker-step-prob (cadr x))))
                       nil
                       (list (walker-step-params (car x)))))
                 (steps)))
        (:forward-steps
         (mapcon (lambda (x)
                   (when (cadr x)
                     (if (<= (walker-step-prob (car x)) (walker-step-prob (cadr x)))
                         nil
                         (list (walker-step-params (car x))))))
                 (steps)))
        (:most-likely-step
         (reduce (lambda (x y) (if (> (walker-step-prob x) (walker-step-prob y)) x y)) (steps)))
        (:most-likely-params (walker-step-params (walker-most-likely-step walker chain)))
        (:median-params
         (let ((s (steps)))
           (loop for k in (walker-param-keys walker)
                 append (list k (median (mapcar (lambda (st) (getf (walker-step-params st) k)) s))))))
        (:acceptance
         (cffi:with-foreign-object (out :double (walker-n-chains walker))
           (with-c-call (check (%mhx-get-acceptance e (max 1 take) out)))
           (/ (round (* (cffi:mem-aref out :double chain) take)) take)))
        (:l-matrix
         (cffi:with-foreign-objects ((lm :double (* d d)) (st :int) (nf :int))
           (with-c-call (check (%mhx-get-proposal-factor e chain (max 1 take) lm st nf)))
           (case (cffi:mem-ref st :int)
             (0 (let ((a (make-array (list d d) :element-type 'double-float)))
                  (dotimes (i d a)
                    (dotimes (j d)
                      (setf (aref a i j) (cffi:mem-aref lm :double (+ (* i d) j)))))))
             (1 (error 'division-by-zero :operation 'cholesky-decomp :operands nil))
             (2 (error 'floating-point-invalid-operation :operation 'cholesky-decomp
                                                          :operands nil))
             (t (make-array '(0 0) :element-type 'double-float)))))
        (t (error "walker-get: ~s is host-side post-processing the engine does not serve" get))))))

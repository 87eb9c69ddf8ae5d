;;;; walker.lisp -- walker-create / walker-adaptive-steps / walker-get over libmhx.
;;; Mirrors the reference's surface for the walker-adaptive-steps path; every number is
;;; computed below the C ABI.  Citations M: = mcmc-fitting.lisp of the reference.
(in-package #:mcmc-fitting-amd)

(defvar mfit-walker-estop nil
  "As in the reference (M:860-861): walker-adaptive-steps-full clears it on entry (M:865) and
looks at it between launches of the stepping kernel (about every 50 ms of GPU work, where the
reference looks once per iteration, M:904).  Set it from an interrupt handler or another
thread and the walk stops where it is; (request-stop walker) does the same without the
variable.")

(defstruct walker-step
  (prob most-negative-double-float :type float)
  (params nil :type list))

(defstruct walker
  (engine (cffi:null-pointer))          ; mhx_engine* (single device)
  (group (cffi:null-pointer))           ; mhx_group*  (:devices with more than one entry)
  (ranges nil :type list)               ; group: ((first count engine-pointer) ...) per device
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

(defun clean-data (data number-of-functions)
  "The layouts walker-create accepts for :data (M:807-825): one dataset (x-column y-column), or
a list of them, one per function; columns may be lists or vectors."
  (let ((depth (get-depth data)))
    (when (and depth (= depth 2))       ; a single dataset: wrap it
      (setf data (list data)
            depth 3))
    ;; one dataset whose x elements are vectors - "multiple or linked independent variables",
    ;; mcmc-fitting.lisp:1136-1137: ((#(a1 b1) #(a2 b2) ...) (y1 y2 ...)) for ONE function
    (when (and depth (= depth 3) (= number-of-functions 1) (= (length data) 2)
               (every #'realp (coerce (second data) 'list)))
      (setf data (list data)
            depth 4))
    (unless (and depth (>= depth 3))
      (error "walker-create: :data must be (x-column y-column) or a list of such datasets"))
    (unless (= (length data) number-of-functions)
      (error "walker-create: ~d dataset~:p for ~d function~:p"
             (length data) number-of-functions))
    (loop for dataset in data
          collect (loop for column in dataset collect (coerce column 'list)))))

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

(defun grouped-p (walker)
  (not (cffi:null-pointer-p (walker-group walker))))

(defun engine-of (walker &optional (chain 0))
  "values: the mhx_engine* that owns global chain CHAIN, and the chain's index on that engine"
  (cond ((grouped-p walker)
         (loop for (first count engine) in (walker-ranges walker)
               when (< -1 (- chain first) count)
                 do (return (values engine (- chain first)))
               finally (error "chain ~d outside the walker's ~d chains" chain
                              (walker-n-chains walker))))
        ((cffi:null-pointer-p (walker-engine walker)) (error "walker has been destroyed"))
        (t (values (walker-engine walker) chain))))

(defun all-engines (walker)
  (if (grouped-p walker)
      (mapcar #'third (walker-ranges walker))
      (list (engine-of walker))))

(defun signal-if-trapped (walker)
  "A frozen chain is where the reference would have signalled an unhandled float trap."
  (loop for (first count engine) in (if (grouped-p walker)
                                         (walker-ranges walker)
                                         (list (list 0 (walker-n-chains walker) (engine-of walker))))
        do (cffi:with-foreign-object (st :int32 count)
             (with-c-call (check (%mhx-get-chain-status engine st (cffi:null-pointer))))
             (dotimes (c count)
               (when (= (cffi:mem-aref st :int32 c) +chain-fp-trap+)
                 (error 'floating-point-invalid-operation
                        :operation 'walker-take-step :operands (list :chain (+ first c))))))))

;;; ------------------------------------------------------------------ walker-create
(defun %define-problem (walker set-function set-function-expr set-dataset set-dataset-cols
                        set-bounds set-prior-expr set-likelihood-expr)
  "walker-create's per-function work (M:1138-1147), through the seven setters of one engine or of
a group (each takes the function index k first and applies CHECK itself)."
  (let* ((keys (walker-param-keys walker)) (d (length keys)))
    (loop for fn in (walker-function walker)
          for k from 0
          for ds in (walker-data walker)
          for sg in (walker-data-error walker)
          for lik in (walker-log-liklihood walker)
          for pri in (walker-log-prior walker)
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
                       (funcall set-function-expr k (model-expr fn) cnames cidx (length idx)))
                     (funcall set-function k (model-id fn) cshape (length shape) cidx
                              (length idx))))
               (cffi:with-foreign-objects ((cx :double (max 1 n)) (cy :double (max 1 n))
                                           (cs :double (max 1 n)) (cx1 :double (max 1 n))
                                           (cols :pointer 2))
                 (fill-doubles cy (second ds))
                 (fill-doubles cs sg)
                 (if (and (plusp n) (typep (elt (first ds) 0) 'sequence))
                     ;; a vector-valued x (mcmc-fitting.lisp:1136-1137): its two components as
                     ;; columns, (elt x 0) -> xcol0, (elt x 1) -> xcol1 (expr.lisp)
                     (progn
                       (fill-doubles cx (mapcar (lambda (v) (elt v 0)) (first ds)))
                       (fill-doubles cx1 (mapcar (lambda (v) (elt v 1)) (first ds)))
                       (setf (cffi:mem-aref cols :pointer 0) cx
                             (cffi:mem-aref cols :pointer 1) cx1)
                       (funcall set-dataset-cols k cols 2 cy cs n (likelihood-id lik)))
                     (progn
                       (fill-doubles cx (first ds))
                       (funcall set-dataset k cx cy cs n (likelihood-id lik)))))
               (when (likelihood-spec-p lik)
                 (funcall set-likelihood-expr k (likelihood-spec-expr lik)))
               (let ((bounds (cond ((null pri) nil)
                                   ((eq pri 'log-prior-flat) nil)
                                   ((eq pri #'log-prior-flat) nil)
                                   ((prior-bounds-spec-p pri) (prior-bounds-spec-bounds pri))
                                   (t (error 'mhx-error
                                             :code -5
                                             :message ":log-prior must be log-prior-flat or (prior-bounds ...)")))))
                 (cffi:with-foreign-objects ((bi :int32 (max 1 (length bounds)))
                                             (lo :double (max 1 (length bounds)))
                                             (hi :double (max 1 (length bounds))))
                   (fill-int32s bi (mapcar (lambda (b) (or (position (first b) keys) -1)) bounds))
                   (fill-doubles lo (mapcar #'second bounds))
                   (fill-doubles hi (mapcar #'third bounds))
                   (funcall set-bounds k bi lo hi (length bounds)))
                 (when (and (prior-bounds-spec-p pri) (prior-bounds-spec-body-expr pri))
                   (cffi:with-foreign-object (gi :int32 (max 1 d))
                     (fill-int32s gi (loop for i below d collect i))
                     (with-c-strings (cnames (mapcar #'mangle-symbol keys))
                       (funcall set-prior-expr k (prior-bounds-spec-body-expr pri)
                                cnames gi d)))))))))

(defun walker-create (&key function data params data-error log-liklihood log-prior param-bounds
                        (n-chains 1) (device 0) devices (seed 0) (chain-offset 0)
                        (history-capacity 0) (pooled nil))
  "(walker-create &key function data params data-error log-liklihood log-prior param-bounds)
M:1132-1163.  :function takes model designators (models.lisp); everything else as the
reference, single items or lists with one item per function.  N-CHAINS > 1 makes a walker
set that steps as one batch; :DEVICES '(0 1 ...) spreads the set over several GPUs from this
one Lisp image (mhx_group_*: contiguous chain ranges, every GPU busy at once); :POOLED t pools
the adaptive proposal covariance over all chains (one RCCL all-reduce per 200 steps)."
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
         (devices (or devices (list device)))
         (grouped (> (length devices) 1))
         (walker (make-walker :function function :param-keys keys :data data
                              :data-error data-error :log-liklihood liks :log-prior pris
                              :n-chains n-chains :n-params d))
         (ok nil))
    (unless (every #'model-p function)
      (error 'mhx-error :code -5 :message ":function must be a model designator (see models.lisp)"))
    (cffi:with-foreign-objects ((cfg '(:struct mhx-config)) (out :pointer)
                                (devs :int32 (length devices)))
      (dotimes (i (cffi:foreign-type-size '(:struct mhx-config)))
        (setf (cffi:mem-aref cfg :uint8 i) 0))
      (setf (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-chains) n-chains
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-params) d
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'n-functions) k-fns
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'device) (first devices)
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'adapt-mode) (if pooled 1 0)
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'seed) seed
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'chain-offset) chain-offset
            (cffi:foreign-slot-value cfg '(:struct mhx-config) 'history-capacity) history-capacity)
      (fill-int32s devs devices)
      (cond (grouped
             (with-c-call (check (%mhx-group-create cfg devs (length devices) out)))
             (setf (walker-group walker) (cffi:mem-ref out :pointer))
             (setf (walker-ranges walker)
                   (loop for i below (%mhx-group-size (walker-group walker))
                         collect (cffi:with-foreign-objects ((first :int64) (count :int64))
                                   (check (%mhx-group-chain-range (walker-group walker) i
                                                                  first count))
                                   (list (cffi:mem-ref first :int64) (cffi:mem-ref count :int64)
                                         (%mhx-group-engine (walker-group walker) i))))))
            (t
             (with-c-call (check (%mhx-create cfg out)))
             (setf (walker-engine walker) (cffi:mem-ref out :pointer)))))
    (unwind-protect
         (let ((e (walker-engine walker)) (g (walker-group walker)))
           (if grouped
               (%define-problem
                walker
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-function g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-function-expr g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-dataset g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-dataset-cols g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-bounds g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-prior-expr g k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-group-set-likelihood-expr g k a)))))
               (%define-problem
                walker
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-function e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-function-expr e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-dataset e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-dataset-cols e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-bounds e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-prior-expr e k a))))
                (lambda (k &rest a) (with-c-call (check (apply #'%mhx-set-likelihood-expr e k a))))))
           (cffi:with-foreign-object (th :double d)
             (fill-doubles th values)
             (with-c-call (check (if grouped
                                     (%mhx-group-init-chains g th 1)
                                     (%mhx-init-chains e th 1)))))
           (setf ok t))
      (unless ok (walker-destroy walker)))
    (signal-if-trapped walker)
    walker))

(defun walker-destroy (walker)
  (cond ((grouped-p walker)
         (with-c-call (%mhx-group-destroy (walker-group walker)))
         (setf (walker-group walker) (cffi:null-pointer)
               (walker-ranges walker) nil))
        ((not (cffi:null-pointer-p (walker-engine walker)))
         (with-c-call (%mhx-destroy (walker-engine walker)))
         (setf (walker-engine walker) (cffi:null-pointer))))
  nil)

(defun request-stop (walker)
  "What (setf mfit-walker-estop t) does in the reference (M:860-861, polled at M:904)."
  (if (grouped-p walker)
      (with-c-call (check (%mhx-group-request-stop (walker-group walker))))
      (with-c-call (check (%mhx-request-stop (engine-of walker))))))

;;; ------------------------------------------------------------------ state read-back
(defun %state (walker chain)
  "values: theta prob best-theta best-prob length age  of chain CHAIN (d doubles cross the ABI,
not the state of every chain)"
  (multiple-value-bind (e c) (engine-of walker chain)
    (let ((d (walker-n-params walker)))
      (cffi:with-foreign-objects ((th :double d) (lp :double) (bt :double d) (bl :double)
                                  (ln :int64) (ag :int64))
        (with-c-call (check (%mhx-get-chain e c th lp bt bl ln ag)))
        (values (read-doubles th d) (cffi:mem-ref lp :double)
                (read-doubles bt d) (cffi:mem-ref bl :double)
                (cffi:mem-ref ln :int64) (cffi:mem-ref ag :int64))))))

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
  (loop for e in (all-engines walker)
        for n in (if (grouped-p walker)
                     (mapcar #'second (walker-ranges walker))
                     (list (walker-n-chains walker)))
        append (cffi:with-foreign-object (st :int32 n)
                 (with-c-call (check (%mhx-get-chain-status e st (cffi:null-pointer))))
                 (loop for c below n collect (cffi:mem-aref st :int32 c)))))

(defun walker-kernel-name (walker)
  "Which kernels serve this walker's problem, e.g. \"w8/lorder_normal\" or
\"w8/rtc[expr:normal] split x24\" (mhx_kernel_name)."
  (with-c-call (%mhx-kernel-name (engine-of walker))))

(defun %trace (walker chain take)
  "newest-first list of walker-steps, (walker-get :get :steps :take take)"
  (multiple-value-bind (e c) (engine-of walker chain)
    (let* ((d (walker-n-params walker))
           (take (max 1 take)))
      (cffi:with-foreign-objects ((pr :double take) (th :double (* take d)) (n-out :int))
        (with-c-call (check (%mhx-get-trace e c take pr th n-out)))
        ;; the reference keeps every step of a walk (M:549), the engine the newest
        ;; history-capacity in its device ring: a window that reaches past the ring is answered
        ;; with what is there, and says so
        (when (< (cffi:mem-ref n-out :int) take)
          (warn "walker-get :take ~d: the device history ring holds the newest ~d steps of this walk; create the walker with :history-capacity >= the walk's length to keep them all"
                take (cffi:mem-ref n-out :int)))
        (loop for s below (cffi:mem-ref n-out :int)
              collect (make-walker-step
                       :prob (cffi:mem-aref pr :double s)
                       :params (%plist walker (read-doubles (cffi:inc-pointer th (* 8 s d)) d))))))))

(defun walker-walk (walker &optional (chain 0))
  (%trace walker chain (walker-length walker chain)))

(defun walker-modify (walker &key modify burn-number keep-number &allow-other-keys)
  "M:547-580.  :add-step happens on the device inside walker-take-step; :add-walks is unused
by the reference itself (M:556 discards the nconc)."
  (flet ((each (action n)
           (dolist (e (all-engines walker))
             (with-c-call (check (%mhx-walker-modify e action n))))))
    (ecase modify
      (:burn-walks (each 0 burn-number))
      (:keep-walks (each 1 keep-number))
      (:reset (each 2 0) walker)
      (:reset-to-most-likely (each 3 0) walker)
      (:delete (walker-destroy walker)))))

;;; ------------------------------------------------------------------ stepping
(defun %fill-matrix (lm l-matrix d)
  (dotimes (i d lm)
    (dotimes (j d)
      (setf (cffi:mem-aref lm :double (+ (* i d) j)) (coerce (aref l-matrix i j) 'double-float)))))

(defun %with-matrix (l-matrix d fn)
  (cffi:with-foreign-object (lm :double (* d d))
    (funcall fn (%fill-matrix lm l-matrix d))))

(defun diagonal-covariance (values)
  "a d x d matrix with VALUES on its diagonal (M:605-611)"
  (let* ((d (length values))
         (a (make-array (list d d) :element-type 'double-float :initial-element 0d0)))
    (loop for v in values
          for i from 0
          do (setf (aref a i i) (coerce v 'double-float)))
    a))

(defun walker-adaptive-steps-full (walker &key (n 100000) (temperature 1d3)
                                            (auto :prob-settle)
                                            (sampling-optimization :covariance)
                                            max-walker-length l-matrix)
  "M:862-942.  The do loop runs on the GPU in launches of about 50 ms; between launches this
function looks at MFIT-WALKER-ESTOP, as the reference does at the top of every iteration
(M:904), and raises the device's stop flag when it is set.  :auto :slope-settle and
:sampling-optimization :best-value are outside the accelerated path and signal MHX-ERROR."
  (unless (eq sampling-optimization :covariance)
    (error 'mhx-error :code -5 :message ":best-value is outside the accelerated path"))
  (when (eq auto :slope-settle)
    (error 'mhx-error :code -5 :message ":slope-settle is outside the accelerated path"))
  (setf mfit-walker-estop nil)          ; M:865
  (let* ((d (walker-n-params walker))
         (grouped (grouped-p walker))
         (g (walker-group walker))
         (e (engine-of walker)))
    (cffi:with-foreign-objects ((o '(:struct mhx-run-opts)) (lm :double (* d d))
                                (running :int64) (avg :double) (launches :uint64)
                                (total :double))
      (%mhx-run-opts-default o)
      (setf (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'n) (floor n)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'temperature)
            (coerce temperature 'double-float)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'auto-mode) (if auto 1 0)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'max-walker-length)
            (if max-walker-length (floor max-walker-length) 0)
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'l-matrix-per-chain) 0
            (cffi:foreign-slot-value o '(:struct mhx-run-opts) 'l-matrix)
            (if l-matrix (%fill-matrix lm l-matrix d) (cffi:null-pointer)))
      (with-c-call (check (if grouped
                              (%mhx-group-adaptive-begin g o)
                              (%mhx-adaptive-begin e o))))
      ;; M:902-942 in launches; the launch length follows the measured time per iteration
      (let ((chunk 16) (stop-sent nil))
        (loop
          (when (and mfit-walker-estop (not stop-sent))
            (request-stop walker)
            (setf stop-sent t))
          (with-c-call (check (%mhx-kernel-timing e 1 avg launches total)))
          (with-c-call (check (if grouped
                                  (%mhx-group-adaptive-advance g chunk running)
                                  (%mhx-adaptive-advance e chunk running))))
          (when (zerop (cffi:mem-ref running :int64)) (return))
          (with-c-call (check (%mhx-kernel-timing e 0 avg launches total)))
          (let ((ms (cffi:mem-ref total :double)))
            (when (> ms 0d0)
              (setf chunk (max 8 (min 65536 (floor (* chunk 50d0) ms))))))))))
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

(defun %scaled-diagonal (plist)
  "(diagonal-covariance (plist-values (scale-plist 1e-2 plist))) of M:851 / M:1074.  The 1e-2 is
a SINGLE float there, so every entry is (* 1e-2 value) with the single widened to a double."
  (diagonal-covariance (loop for (nil v) on plist by #'cddr collect (* 1e-2 v))))

(defun walker-many-steps (walker n &optional l-matrix)
  "(walker-many-steps the-walker n &optional l-matrix) M:849-853: N steps with a constant
l-matrix, temperature 1; without one, 1e-2 of the median parameters on the diagonal (M:851)."
  (let ((l-matrix (or l-matrix
                      (%scaled-diagonal (walker-get walker :get :median-params)))))
    (%with-matrix l-matrix (walker-n-params walker)
                  (lambda (lm)
                    (dolist (e (all-engines walker))
                      (with-c-call (check (%mhx-many-steps e n lm 0)))))))
  (signal-if-trapped walker)
  nil)

(defun walker-take-step (walker &key l-matrix (temperature 1))
  "(walker-take-step walker &key l-matrix (temperature 1)) M:1072-1095: one step of every chain
with the device's own randomness (the reference draws its own too, M:687 / M:1092); without an
l-matrix, 1e-2 of the most likely parameters on the diagonal (M:1074)."
  (let ((l-matrix (or l-matrix
                      (%scaled-diagonal (walker-get walker :get :most-likely-params :take 1000)))))
    (%with-matrix l-matrix (walker-n-params walker)
                  (lambda (lm)
                    (dolist (e (all-engines walker))
                      (with-c-call
                        (check (%mhx-take-step e lm 0 (coerce temperature 'double-float))))))))
  (signal-if-trapped walker)
  walker)

(defun walker-take-step-injected (walker &key l-matrix (temperature 1) z u)
  "The parity hook: M:1072-1095 for a single-chain walker with the CALLER's randomness, Z = the d
numbers alexandria:gaussian-random would return (M:687), U = (random 1.0d0) (M:1092).
Returns true when the proposal was taken."
  (let* ((e (engine-of walker)) (d (walker-n-params walker)))
    (unless (= (walker-n-chains walker) 1)
      (error "walker-take-step-injected: single-chain walkers only"))
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

(defun %covariance-of-plists (plists keys)
  "Population covariance of the parameter vectors in PLISTS (what lplist-covariance, M:614-643,
gives for :covariance-matrix): averages first, then for every matrix entry the sum over the
vectors of (x_i - mean_i)(x_j - mean_j) / n, the division inside the sum."
  (let* ((n (length plists))
         (d (length keys))
         (rows (mapcar (lambda (pl) (mapcar (lambda (k) (getf pl k)) keys)) plists))
         (means (loop for i below d
                      collect (/ (loop for row in rows sum (nth i row)) n)))
         (cov (make-array (list d d) :element-type 'double-float :initial-element 0d0)))
    (dotimes (i d cov)
      (dotimes (j d)
        (let ((mi (nth i means)) (mj (nth j means)) (acc 0d0))
          (dolist (row rows)
            (incf acc (/ (* (- (nth i row) mi) (- (nth j row) mj)) n)))
          (setf (aref cov i j) acc))))))

(defun walker-get (walker &key (get :steps) take param (chain 0))
  (let* ((len (walker-length walker chain))
         (take (if take (min len take) len))
         (d (walker-n-params walker))
         (keys (walker-param-keys walker)))
    (flet ((steps () (%trace walker chain take)))
      (case get
        (:steps (steps))
        (:log-liklihoods (mapcar #'walker-step-prob (steps)))
        (:params (mapcar #'walker-step-params (steps)))
        (:param (mapcar (lambda (s) (getf (walker-step-params s) param)) (steps)))
        (:unique-steps
         ;; the parameters of every step whose prob differs from that of the step before it in
         ;; time (the walk is newest first; the oldest step always counts), M:492-496
         (loop for (newer older) on (steps)
               unless (and older (equal (walker-step-prob newer) (walker-step-prob older)))
                 collect (walker-step-params newer)))
        (:forward-steps
         ;; the parameters of every step that improved on the one before it, M:497-502
         (loop for (newer older) on (steps)
               when (and older (> (walker-step-prob newer) (walker-step-prob older)))
                 collect (walker-step-params newer)))
        (:most-likely-step
         (reduce (lambda (x y) (if (> (walker-step-prob x) (walker-step-prob y)) x y)) (steps)))
        (:most-likely-params (walker-step-params (walker-most-likely-step walker chain)))
        (:median-params
         (let ((s (steps)))
           (loop for k in keys
                 append (list k (median (mapcar (lambda (st) (getf (walker-step-params st) k)) s))))))
        (:acceptance
         (multiple-value-bind (e c) (engine-of walker chain)
           (let ((n (if (grouped-p walker)
                        (second (find e (walker-ranges walker) :key #'third :test #'cffi:pointer-eq))
                        (walker-n-chains walker))))
             (cffi:with-foreign-object (out :double n)
               (with-c-call (check (%mhx-get-acceptance e (max 1 take) out)))
               (/ (round (* (cffi:mem-aref out :double c) take)) take)))))
        (:l-matrix
         (multiple-value-bind (e c) (engine-of walker chain)
           (cffi:with-foreign-objects ((lm :double (* d d)) (st :int) (nf :int))
             (with-c-call (check (%mhx-get-proposal-factor e c (max 1 take) lm st nf)))
             (case (cffi:mem-ref st :int)
               (0 (let ((a (make-array (list d d) :element-type 'double-float)))
                    (dotimes (i d a)
                      (dotimes (j d)
                        (setf (aref a i j) (cffi:mem-aref lm :double (+ (* i d) j)))))))
               (1 (error 'division-by-zero :operation 'cholesky-decomp :operands nil))
               (2 (error 'floating-point-invalid-operation :operation 'cholesky-decomp
                                                            :operands nil))
               (t (make-array '(0 0) :element-type 'double-float))))))
        (:stddev-params
         ;; M:525-539: the diagonal of the l-matrix (0d0 while the walk is shorter than 10);
         ;; the l-matrix itself is the second value
         (if (< len 10)
             (loop for k in keys append (list k 0d0))
             (let ((l (walker-get walker :get :l-matrix :take take :chain chain)))
               (values (loop for k in keys
                             for i from 0
                             append (list k (aref l i i)))
                       l))))
        (:covariance-matrix             ; M:541
         (%covariance-of-plists (walker-get walker :get :unique-steps :take take :chain chain)
                                keys))
        (t (error "walker-get: unknown :get ~s" get))))))

;;; ------------------------------------------------------------------ save / load M:971-1001
;;; The plist the reference's (commented) walker-construct-print-list builds, written and read
;;; under with-standard-io-syntax; functions are only NAMED in the file, so walker-load wants
;;; the :function / :log-liklihood / :log-prior designators again, exactly as there.
(defun %designator-name (x)
  (cond ((null x) nil)
        ((symbolp x) x)
        ((model-p x) (or (model-expr x) (list :model (model-id x) (model-shape x))))
        ((likelihood-spec-p x) (likelihood-spec-expr x))
        ((prior-bounds-spec-p x) (list :prior-bounds (prior-bounds-spec-bounds x)))
        (t (princ-to-string x))))

(defun walker-save (walker filename &optional take (chain 0))
  "(walker-save walker filename &optional take): the newest TAKE steps (all of them by default)
of chain CHAIN with the data they were walked on."
  (let ((form (list :fn (mapcar #'%designator-name (walker-function walker))
                    :data (walker-data walker)
                    :param-keys (walker-param-keys walker)
                    :stddev (walker-data-error walker)
                    :log-liklihood (mapcar #'%designator-name (walker-log-liklihood walker))
                    :log-prior (mapcar #'%designator-name (walker-log-prior walker))
                    :walker (mapcar (lambda (s)
                                      (list :prob (walker-step-prob s)
                                            :params (walker-step-params s)))
                                    (walker-get walker :get :steps :take take :chain chain)))))
    (with-open-file (out filename :direction :output :if-exists :supersede)
      (with-standard-io-syntax
        (let ((*package* (find-package :keyword)))
          (write form :stream out)
          (terpri out))))
    nil))

(defun walker-load (filename &key function log-liklihood log-prior quiet (device 0) (seed 0))
  "(walker-load filename &key function log-liklihood log-prior quiet): without the designators,
print what the file recommends and return nil; with :function (and optionally the others),
rebuild the walker on the GPU and restore the saved walk (mhx_set_history)."
  (let* ((full (with-open-file (in filename :direction :input)
                 (with-standard-io-syntax
                   (let ((*package* (find-package :keyword)) (*read-eval* nil))
                     (read in)))))
         (data (getf full :data))
         (stddev (getf full :stddev))
         (keys (getf full :param-keys))
         (walks (getf full :walker)))
    (unless quiet
      (format t "*Recommendations*~%function: ~s~%log-liklihood: ~s~%log-prior: ~s~%"
              (getf full :fn) (getf full :log-liklihood) (getf full :log-prior)))
    (when function
      (let* ((newest (first walks))
             (walker (walker-create :function function :data data
                                    :params (getf newest :params) :data-error stddev
                                    :log-liklihood log-liklihood :log-prior log-prior
                                    :device device :seed seed))
             (n (length walks))
             (d (length keys)))
        (cffi:with-foreign-objects ((pr :double n) (th :double (* n d)))
          (loop for w in walks
                for s from 0
                do (setf (cffi:mem-aref pr :double s) (coerce (getf w :prob) 'double-float))
                   (loop for k in keys
                         for j from 0
                         do (setf (cffi:mem-aref th :double (+ (* s d) j))
                                  (coerce (getf (getf w :params) k) 'double-float))))
          (with-c-call (check (%mhx-set-history (engine-of walker) 0 pr th n))))
        walker))))

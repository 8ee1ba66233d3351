classdef TinyMPC < handle
    % TinyMPC  MATLAB front end of the MI355X (HIP) TinyMPC solver.
    %
    % Drop-in for the reference class of the same name: same methods, argument meaning, defaults and
    % error identifiers, so existing scripts (solver = TinyMPC(); solver.setup(A,B,Q,R,N,'rho',1);
    % solver.set_x0(x0); solver.solve(); sol = solver.get_solution();) run unchanged. Every method
    % forwards to the MEX function tinympc_matlab(verb, ...) built from tinympc_matlab_mex.cpp, which
    % calls the C ABI of libtinympc_hip.so (include/tinympc_hip.h). The ADMM solve itself runs as one
    % HIP kernel on the GPU; this file only shapes arguments.
    %
    % codegen / codegen_with_sensitivity write the data files from the cache on the device; adaptive_rho runs the
    % old core's adaptive-rho loop on the device (per instance).
    % compute_cache_terms / solve_lqr / compute_sensitivity_autograd run on the device (extra MEX verbs).
    % get_stats and get_cache are additions.

    properties
        nx = 0; nu = 0; N = 0;
        A = []; B = []; Q = []; R = [];
        rho = 1.0;
        is_setup = false;
        settings = struct();
        x_min = []; x_max = []; u_min = []; u_max = [];
        dK = []; dP = []; dC1 = []; dC2 = [];
    end

    properties (Constant, Access = private)
        BOUND_FILL = 1e17;   % stands for "no bound" on an unspecified side
        SETTING_ORDER = {'abs_pri_tol', 'abs_dua_tol', 'max_iter', 'check_termination', ...
            'en_state_bound', 'en_input_bound', 'en_state_soc', 'en_input_soc', ...
            'en_state_linear', 'en_input_linear', 'adaptive_rho', 'adaptive_rho_min', ...
            'adaptive_rho_max', 'adaptive_rho_enable_clipping'};   % argument order of the update_settings verb
    end

    methods
        function obj = TinyMPC()
            s = struct();
            s.abs_pri_tol = 1e-4;  s.abs_dua_tol = 1e-4;
            s.max_iter = 100;      s.check_termination = 1;
            s.en_state_bound = false;  s.en_input_bound = false;
            s.en_state_soc = false;    s.en_input_soc = false;
            s.en_state_linear = false; s.en_input_linear = false;
            s.adaptive_rho = false; s.adaptive_rho_min = 0.1; s.adaptive_rho_max = 10.0;
            s.adaptive_rho_enable_clipping = true;
            obj.settings = s;
        end

        function setup(obj, A, B, Q, R, N, varargin)
            % setup(A, B, Q, R, N, 'rho', 1.0, 'fdyn', f, 'max_iter', 100, 'abs_pri_tol', 1e-4, ..., 'verbose', false)
            assert(size(A, 1) == size(A, 2), 'A must be square');
            assert(size(A, 1) == size(B, 1), 'A and B row dimensions must match');
            assert(size(Q, 1) == size(A, 1), 'Q must match A dimensions');
            assert(size(R, 1) == size(B, 2), 'R must match B column dimension');
            assert(N >= 2, 'N must be >= 2');
            [obj.nx, obj.nu, obj.N] = deal(size(A, 1), size(B, 2), N);
            [obj.A, obj.B, obj.Q, obj.R] = deal(A, B, Q, R);

            opt = struct('rho', 1.0, 'fdyn', [], 'verbose', false);
            tunable = {'abs_pri_tol', 'abs_dua_tol', 'max_iter', 'check_termination', 'adaptive_rho', ...
                       'adaptive_rho_min', 'adaptive_rho_max', 'adaptive_rho_enable_clipping'};
            for k = 1:2:numel(varargin) - 1          % name/value pairs; names we do not know are ignored
                name = varargin{k};
                if isfield(opt, name)
                    opt.(name) = varargin{k + 1};
                elseif any(strcmp(name, tunable))
                    obj.settings.(name) = varargin{k + 1};
                end
            end
            obj.rho = opt.rho;
            % bounds become active only through set_bound_constraints
            obj.settings.en_state_bound = false;
            obj.settings.en_input_bound = false;
            fdyn = opt.fdyn;
            if isempty(fdyn), fdyn = zeros(obj.nx, 1); end

            status = tinympc_matlab('setup', A, B, fdyn(:), Q, R, obj.rho, obj.nx, obj.nu, obj.N, opt.verbose);
            if status ~= 0
                error('TinyMPC:SetupFailed', 'Setup failed with status %d', status);
            end
            obj.is_setup = true;
            obj.push_settings();
            if opt.verbose
                fprintf('TinyMPC solver setup successful (nx=%d, nu=%d, N=%d)\n', obj.nx, obj.nu, obj.N);
            end
        end

        function set_x0(obj, x0)
            obj.require_setup();
            tinympc_matlab('set_x0', x0(:), false);
        end

        function set_x_ref(obj, x_ref)
            obj.require_setup();
            tinympc_matlab('set_x_ref', TinyMPC.spread(x_ref, obj.nx, obj.N, []), false);
        end

        function set_u_ref(obj, u_ref)
            obj.require_setup();
            tinympc_matlab('set_u_ref', TinyMPC.spread(u_ref, obj.nu, obj.N - 1, []), false);
        end

        function update_settings(obj, varargin)
            obj.require_setup();
            for k = 1:2:numel(varargin) - 1
                if isfield(obj.settings, varargin{k})
                    obj.settings.(varargin{k}) = varargin{k + 1};
                end
            end
            obj.push_settings();
        end

        function status = solve(obj)
            % Always returns 0; use get_stats() for iterations / convergence.
            obj.require_setup();
            tinympc_matlab('solve', false);
            status = 0;
        end

        function solution = get_solution(obj)
            obj.require_setup();
            [states, controls] = tinympc_matlab('get_solution', false);
            solution = struct('states', states, 'controls', controls);
        end

        function stats = get_stats(obj)
            obj.require_setup();
            [it, st, pri_x, pri_u] = tinympc_matlab('get_stats', false);
            stats = struct('iter', it, 'status', st, 'primal_residual_state', pri_x, 'primal_residual_input', pri_u);
        end

        function set_bound_constraints(obj, x_min, x_max, u_min, u_max)
            obj.require_setup();
            F = TinyMPC.BOUND_FILL;
            obj.x_min = TinyMPC.spread(x_min, obj.nx, obj.N, -F);
            obj.x_max = TinyMPC.spread(x_max, obj.nx, obj.N, +F);
            obj.u_min = TinyMPC.spread(u_min, obj.nu, obj.N - 1, -F);
            obj.u_max = TinyMPC.spread(u_max, obj.nu, obj.N - 1, +F);
            tinympc_matlab('set_bound_constraints', obj.x_min, obj.x_max, obj.u_min, obj.u_max, false);
            obj.settings.en_state_bound = true;
            obj.settings.en_input_bound = true;
            obj.push_settings();
        end

        function set_linear_constraints(obj, Alin_x, blin_x, Alin_u, blin_u)
            % Alin_x * x <= blin_x and Alin_u * u <= blin_u at every knot point.
            obj.require_setup();
            tinympc_matlab('set_linear_constraints', Alin_x, blin_x, Alin_u, blin_u, false);
            obj.settings.en_state_linear = ~isempty(Alin_x) && ~isempty(blin_x);
            obj.settings.en_input_linear = ~isempty(Alin_u) && ~isempty(blin_u);
            if obj.settings.en_state_linear || obj.settings.en_input_linear
                obj.push_settings();
            end
        end

        function set_cone_constraints(obj, Acx, qcx, cx, Acu, qcu, cu)
            % Second-order cones, states first: ||s(Ac+1 : Ac+qc-1)|| <= c * s(Ac+qc) with 0-based start Ac.
            obj.require_setup();
            if ~isempty(Acx), Acx = int32(Acx(:)); qcx = int32(qcx(:)); cx = double(cx(:)); end
            if ~isempty(Acu), Acu = int32(Acu(:)); qcu = int32(qcu(:)); cu = double(cu(:)); end
            tinympc_matlab('set_cone_constraints', Acx, qcx, cx, Acu, qcu, cu, false);
            obj.settings.en_state_soc = ~isempty(Acx) && ~isempty(qcx) && ~isempty(cx);
            obj.settings.en_input_soc = ~isempty(Acu) && ~isempty(qcu) && ~isempty(cu);
            if obj.settings.en_state_soc || obj.settings.en_input_soc
                obj.push_settings();
            end
        end

        function set_equality_constraints(obj, Aeq_x, beq_x, Aeq_u, beq_u)
            % Aeq * s == beq, posed as the pair of inequalities [Aeq; -Aeq] * s <= [beq; -beq].
            obj.require_setup();
            [Ax, bx] = TinyMPC.two_sided(Aeq_x, beq_x);
            [Au, bu] = TinyMPC.two_sided(Aeq_u, beq_u);
            obj.set_linear_constraints(Ax, bx, Au, bu);
        end

        function set_cache_terms(obj, Kinf, Pinf, Quu_inv, AmBKt)
            obj.require_setup();
            tinympc_matlab('set_cache_terms', Kinf, Pinf, Quu_inv, AmBKt, false);
        end

        function set_sensitivity_matrices(obj, dK, dP, dC1, dC2)
            obj.require_setup();
            obj.check_sensitivity_shapes(dK, dP, dC1, dC2);
            [obj.dK, obj.dP, obj.dC1, obj.dC2] = deal(dK, dP, dC1, dC2);
            tinympc_matlab('set_sensitivity_matrices', dK, dP, dC1, dC2, false);
        end

        function codegen(obj, output_dir)
            obj.require_setup();
            status = tinympc_matlab('codegen', output_dir, false);
            if status ~= 0
                error('TinyMPC:CodegenFailed', 'Code generation failed with status: %d', status);
            end
            TinyMPC.place_solver_sources(output_dir);
            fprintf('Code generation completed successfully in: %s\n', output_dir);
        end

        function codegen_with_sensitivity(obj, output_dir, dK, dP, dC1, dC2)
            obj.require_setup();
            obj.set_sensitivity_matrices(dK, dP, dC1, dC2);
            status = tinympc_matlab('codegen_with_sensitivity', output_dir, dK, dP, dC1, dC2, false);
            if status ~= 0
                error('TinyMPC:CodegenWithSensitivityFailed', ...
                      'Code generation with sensitivity failed with status: %d', status);
            end
            TinyMPC.place_solver_sources(output_dir);
            fprintf('Code generation with sensitivity matrices completed successfully in: %s\n', output_dir);
        end

        function [Kinf, Pinf, Quu_inv, AmBKt] = compute_cache_terms(obj)
            % Riccati recursion of the class (full Q and R, rho added once), run on the device.
            obj.require_setup();
            [Kinf, Pinf, Quu_inv, AmBKt] = tinympc_matlab('compute_cache_terms', false);
        end

        function [K, P, C1, C2] = solve_lqr(obj, rho_val)
            % Stabilising DARE solution for Q + rho_val*I, R + rho_val*I, on the device (u = -K*x).
            obj.require_setup();
            [K, P, C1, C2] = tinympc_matlab('solve_lqr', rho_val, false);
        end

        function [dK, dP, dC1, dC2] = compute_sensitivity_autograd(obj)
            % Forward differences of solve_lqr in rho (h = 1e-6), on the device.
            obj.require_setup();
            [dK, dP, dC1, dC2] = tinympc_matlab('compute_sensitivity', false);
        end

        % --- closed-loop session (extension): the solve kernel stays resident between ticks, so a tick costs
        % no kernel launch and no stream synchronisation. Drop-in for the loop body
        %   solver.set_x0(x); solver.solve(); sol = solver.get_solution(); u = sol.controls(:,1);
        % as   u = solver.session_step(x);   between session_begin() and session_end(). get_solution / get_stats
        % keep working inside a session; any other verb ends it.
        function prepare(obj)
            % Choose (and, if needed, specialise) the solve kernel for the current constraints / settings now
            % instead of at the first solve that needs it.
            obj.require_setup();
            tinympc_matlab('prepare');
        end

        function session_begin(obj)
            obj.require_setup();
            tinympc_matlab('session_begin');
        end

        function u0 = session_step(obj, x0)
            obj.require_setup();
            u0 = tinympc_matlab('session_step', double(x0(:)));
        end

        function session_end(obj)
            obj.require_setup();
            tinympc_matlab('session_end');
        end

        % --- resident solves (extension, off by default): after solver.set_resident(true) the usual loop
        %     solver.set_x0(x); solver.solve(); sol = solver.get_solution();
        % runs on the resident kernel instead of one launch per solve (same results bit for bit, a 2.5x
        % shorter tick). The kernel occupies one compute unit while it is resident and leaves after 2 s
        % without a solve; other GPU code of the MATLAB process that synchronises the whole device waits
        % for it that long -- switch it off (false) around such code.
        function set_resident(obj, tf)
            obj.require_setup();
            tinympc_matlab('set_resident', double(logical(tf)));
        end

        function reset(obj)
            if obj.is_setup
                tinympc_matlab('reset', false);
                obj.is_setup = false;
            end
        end
    end

    methods (Access = private)
        function require_setup(obj)
            if ~obj.is_setup
                error('TinyMPC:NotSetup', 'Solver not setup. Call setup() first.');
            end
        end

        function push_settings(obj)
            vals = cellfun(@(f) obj.settings.(f), TinyMPC.SETTING_ORDER, 'UniformOutput', false);
            tinympc_matlab('update_settings', vals{:}, false);
        end

        function check_sensitivity_shapes(obj, dK, dP, dC1, dC2)
            assert(isequal(size(dK), [obj.nu, obj.nx]), 'dK must be nu x nx');
            assert(isequal(size(dP), [obj.nx, obj.nx]), 'dP must be nx x nx');
            assert(isequal(size(dC1), [obj.nu, obj.nu]), 'dC1 must be nu x nu');
            assert(isequal(size(dC2), [obj.nx, obj.nx]), 'dC2 must be nx x nx');
        end
    end

    methods (Static, Access = private)
        function place_solver_sources(output_dir)
            % What the reference class does after the MEX call: put the embedded solver's sources (the
            % codegen_src tree of a TinyMPC checkout: include/, tinympc/, CMakeLists.txt) beside the generated
            % files and create build/. Looked up in $TINYMPC_CODEGEN_SRC, then next to this file.
            src = getenv('TINYMPC_CODEGEN_SRC');
            if isempty(src)
                src = fullfile(fileparts(mfilename('fullpath')), 'codegen_src');
            end
            if isfolder(src)
                copyfile(src, output_dir);
            else
                warning('TinyMPC:CopyArtifactsError', 'No codegen_src tree found (%s): only the generated files were written.', src);
            end
            if ~isfolder(fullfile(output_dir, 'build'))
                mkdir(fullfile(output_dir, 'build'));
            end
        end

        function out = spread(v, rows, cols, fill)
            % Scalar -> constant matrix; rows-vector (either orientation) -> repeated over the horizon;
            % empty -> `fill` everywhere (bounds only); anything else is taken as already rows x cols.
            if isempty(v) && ~isempty(fill)
                out = fill * ones(rows, cols);
            elseif isscalar(v)
                out = v * ones(rows, cols);
            elseif isvector(v) && numel(v) == rows && ~isequal(size(v), [rows, cols])
                out = repmat(v(:), 1, cols);
            else
                out = v;
            end
        end

        function [A2, b2] = two_sided(Aeq, beq)
            if isempty(Aeq)
                A2 = []; b2 = [];
            else
                beq = beq(:);
                A2 = [Aeq; -Aeq];
                b2 = [beq; -beq];
            end
        end
    end
end
